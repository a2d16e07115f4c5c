#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc passes (tools/pmc_kernels.sh): median over the dispatches of every counter, and a few ratios.
usage: python tools/pmc_table.py gpurun_out/pmc_<tag>"""
import csv, glob, sys
from collections import defaultdict
import numpy as np
src = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals, key=lambda k: -np.median(vals[k].get("SQ_WAVE_CYCLES", [0]))):
    v = {c: float(np.median(x)) for c, x in vals[k].items()}
    if v.get("SQ_WAVE_CYCLES", 0) < 1e5 or "copyBuffer" in k:
        continue
    n = len(next(iter(vals[k].values())))
    print(f"{k[:150]}\n    dispatches per pass {n}; medians per dispatch:")
    print("    " + "  ".join(f"{c} {v[c]:.4g}" for c in sorted(v)))
    d = []
    if v.get("SQ_WAVES"):
        d.append(f"VALU instructions per wavefront {v.get('SQ_INSTS_VALU', 0) / v['SQ_WAVES']:.0f}")
        if "SQ_INSTS_MFMA" in v: d.append(f"matrix instructions per wavefront {v['SQ_INSTS_MFMA'] / v['SQ_WAVES']:.0f}")
        if "SQ_INSTS_LDS" in v: d.append(f"LDS instructions per wavefront {v['SQ_INSTS_LDS'] / v['SQ_WAVES']:.0f}")
    if v.get("SQ_WAVE_CYCLES"):
        for c, nm in (("SQ_ACTIVE_INST_VALU", "VALU active"), ("SQ_WAIT_INST_ANY", "waiting to issue"), ("SQ_WAIT_ANY", "waiting on a counter"), ("SQ_ACTIVE_INST_LDS", "LDS active")):
            if c in v: d.append(f"{nm} {v[c] / v['SQ_WAVE_CYCLES']:.2f} of wave cycles")
    if v.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_MFMA" in v:
        d.append(f"matrix pipe busy {v['SQ_INSTS_MFMA'] * 64.0 / 1024.0 / (v['GRBM_GUI_ACTIVE'] / 8.0):.2f} of the kernel's time (64 cycles per instruction, 1024 SIMDs)")
    if "FETCH_SIZE" in v: d.append(f"HBM fetch {v['FETCH_SIZE'] * 2048 / 1e9:.2f} GB (KB x 2: the gfx950 correction), write {v.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB")
    print("    " + "; ".join(d))
