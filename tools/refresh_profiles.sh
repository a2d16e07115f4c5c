#!/bin/bash
# GPU-box script: re-measures every secondary number DESIGN.md quotes on the CURRENT kernels (one JSON / text file each under
# gpurun_out/$ROUND/extra/), then tools/run_profiles.sh for the two throughput workloads.  Copy the results into profiles/ with
# tools/collect_profiles.py + a plain cp of extra/*.  ROUND defaults to r03.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r06}
X=$ROOT/gpurun_out/$ROUND/extra
mkdir -p "$X"
cd "$ROOT"
run() { name=$1; shift; timeout -k 10 300 "$@" > "$X/$name" 2> "$X/$name.err" || echo "$name failed" >&2; tail -c 400 "$X/$name"; echo; }
# PART=1: the secondary measurements; PART=2: the reference's benchmark grid; PART=3: rocprofv3 trace + PMC passes (default: all three;
# a gpurun call is limited to 20 minutes — run the parts as separate calls)
PART=${PART:-all}
if [[ $PART == all || $PART == 1 ]]; then
run predict_simulate_n1e4_b256.json python3 tools/bench_predict.py
run gradient_sho20.json python3 tools/bench_grad.py
BASIS=DRWCelerite run gradient_drw20.json python3 tools/bench_grad.py
run qpo_mixed_b4096.json python3 tools/bench_qpo.py
run qpo_small_batches.txt python3 tools/bench_qpo_small.py
run per_draw_small_batches.json python3 tools/bench_per_draw_small.py
run sweep_block_emode.txt python3 tools/sweep_block_emode.py
run shift_transform_b4096.json python3 tools/bench_shift.py
run small_batch_latency_sho20.json python3 tools/bench_small_batch.py
BASIS=DRWCelerite run small_batch_latency_drw20.json python3 tools/bench_small_batch.py
run host_api_pcie_inclusive.json python3 tools/bench_host_api.py
run dense_n4096_j40.json python3 tools/bench_dense.py
run dense_batched_launches.txt python3 tools/sweep_dense_streams.py
PIORAN_BENCH_FULL=$X/bench_full.json run bench_default_full_line.json python3 bench.py
fi
if [[ $PART == all || $PART == 2 ]]; then
timeout -k 10 900 python3 tools/bench_grid.py > "$X/grid.json" 2> "$X/grid.json.err" || echo "grid failed" >&2
fi
if [[ $PART == all || $PART == 3 ]]; then
ROUND=$ROUND "$ROOT/tools/run_profiles.sh" > "$X/run_profiles.log" 2>&1
tail -5 "$X/run_profiles.log"
fi
