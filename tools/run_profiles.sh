#!/bin/bash
# GPU-box script: rocprofv3 kernel-trace summary + separate PMC passes of the SAME bench.py command, for the headline
# (SHO-20) and the DRWCelerite-20 workloads.  Outputs under gpurun_out/r02/ (scratch); tools/collect_profiles.py then
# writes the summaries the judge reads into profiles/.  The program sits directly after `--` (no env/bash hop).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r06}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for basis in SHO DRWCelerite; do
  tag=$(echo $basis | tr 'A-Z' 'a-z')
  args="$ROOT/bench.py --no-cpu-baseline --no-secondary --basis $basis"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $args --steps 20 --warmup 5 > $OUT/bench_trace_$tag.json 2> $OUT/trace_$tag.err
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64"; do
    name=$(echo $ctr | tr ' ' '+' | cut -c1-40)
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${tag}_$name -- python3 $args --steps 2 --warmup 1 > $OUT/bench_pmc_${tag}_$name.json 2> $OUT/pmc_${tag}_$name.err
  done
  python3 $args --steps 20 --warmup 5 > $OUT/bench_plain_$tag.json 2> $OUT/bench_plain_$tag.err
done
ls -R $OUT | head -60
