#!/bin/bash
# GPU box: per-launch HBM traffic and L2 hit / miss counts of ONE pioran_dense_nll call (separate --pmc passes over tools/bench_dense.py;
# no trace domains beside the counters).  Output: gpurun_out/r06_dense_pmc_per_launch.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_dense; rm -rf $out; mkdir -p $out
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  CPU=0 YARDSTICK=0 REPS=2 timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/g$i -- python3 tools/bench_dense.py > $out/g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/g$i.log; }
done
python3 - <<'PY' > gpurun_out/r06_dense_pmc_per_launch.txt
import csv, glob
from collections import defaultdict
rows = defaultdict(dict)   # dispatch order within a pass -> counters
names = {}
for g in sorted(glob.glob("gpurun_out/pmc_dense/g*/")):
    fs = glob.glob(g + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    rs = list(csv.DictReader(open(fs[0])))
    # dispatches of dense_* kernels in dispatch order; the LAST call = the last 67 (build x2?, diag0, 64 steps, finish)
    byd = defaultdict(dict)
    for r in rs:
        if "dense_" in r["Kernel_Name"]:
            byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    ids = sorted(byd)
    last_diag0 = max(i for i in ids if "diag0" in names[i])
    seq = [i for i in ids if i >= last_diag0]
    for n, i in enumerate(seq):
        rows[n].update(byd[i]); rows[n]["kernel"] = names[i].split("(")[0][-40:]
print("# launch kernel fetch_MB(x2 gfx950) write_MB tcc_hit tcc_miss hit_rate tcc_req ea_rdreq mfma_insts mfma_busy_cycles gui_active")
for n in sorted(rows):
    r = rows[n]
    hit, miss = r.get("TCC_HIT_sum", 0), r.get("TCC_MISS_sum", 0)
    print(n, r["kernel"], f"{r.get('FETCH_SIZE', 0) * 2048 / 1e6:.2f} {r.get('WRITE_SIZE', 0) * 1024 / 1e6:.2f} {hit:.0f} {miss:.0f} {hit / max(1, hit + miss):.3f} {r.get('TCC_REQ_sum', 0):.0f} {r.get('TCC_EA0_RDREQ_sum', 0):.0f} {r.get('SQ_INSTS_MFMA', 0):.0f} {r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f} {r.get('GRBM_GUI_ACTIVE', 0):.0f}")
PY
rm -rf $out/g*/
head -40 gpurun_out/r06_dense_pmc_per_launch.txt
