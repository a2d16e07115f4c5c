#!/bin/bash
# GPU-box script: where do the scan kernel's wavefronts spend their cycles?  One rocprofv3 --pmc pass (8 SQ counters) of bench.py
# per (basis, batch) [env BASES, BATCHES, COUNTERS]; quad-cycle counters summed over wavefronts.  usage: tools/pmc_waits.sh  -> gpurun_out/pmc_waits/*.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_waits
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for basis in ${BASES:-DRWCelerite SHO}; do
  for B in ${BATCHES:-1024 4096}; do
    tag=${basis}_b$B
    rocprofv3 --pmc ${COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS} \
      --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --basis $basis --batch $B --steps 2 --warmup 1 > $OUT/$tag.json 2> $OUT/$tag.err
    python3 - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys
agg = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "celerite_scan_kernel" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
wc = med.get("SQ_WAVE_CYCLES", 1.0)
print(sys.argv[2], " ".join(f"{k}={v:.4g}({v / wc:.3f})" for k, v in sorted(med.items())))
PY
  done
done
