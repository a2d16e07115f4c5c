#!/bin/bash
# GPU-box script: SQ counters of the 80-row throughput shape (SHO-40, 4096 draws): one rocprofv3 --pmc pass of bench.py --components 40
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_r80
mkdir -p "$OUT"; rm -rf "$OUT"/*
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | cut -c1-24 | tr ' ' '_')
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --components 40 --steps 2 --warmup 1 > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - "$OUT/$tag" <<'PY'
import csv, glob, sys
agg = {}; name = None
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "celerite_scan_kernel" in r["Kernel_Name"]:
            name = r["Kernel_Name"]; agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
print(name); print(" ".join(f"{k}={v:.5g}" for k, v in sorted(med.items())))
PY
done
tail -c 600 $OUT/*.json | head -c 1500
