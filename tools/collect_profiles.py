#!/usr/bin/env python3
"""Copy the round's measured artifacts from gpurun_out/ into profiles/ (tracked).  usage: collect_profiles.py <suffix>
where gpurun_out/prof<suffix>, pmc_fetch<suffix>, pmc_write<suffix>, pmc_sq<suffix> hold the rocprofv3 outputs."""
import csv, glob, json, shutil, sys
from pathlib import Path
root = Path(__file__).resolve().parents[1]
g = root / "gpurun_out"; prof = root / "profiles"
psuf, csuf = sys.argv[1], sys.argv[2]
shutil.copy(glob.glob(str(g / f"prof{psuf}/runc/*kernel_stats.csv"))[0], prof / "r01_bench_sho20_b4096_kernel_stats.csv")
for src, dst in (("bench_r01_sho.json", "r01_bench_sho20_b4096.json"), ("bench_r01_drw.json", "r01_bench_drw20_b4096.json"),
                 ("qpo_r01.json", "r01_qpo_mixed_b4096.json")):
    if (g / src).exists(): shutil.copy(g / src, prof / dst)
kname = [r["Name"] for r in csv.DictReader(open(prof / "r01_bench_sho20_b4096_kernel_stats.csv")) if "celerite_scan" in r["Name"]][0]
out = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "
                  "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*)", "kernel": kname, "workload": "N=1e4, SHO-20 (J=20), B=4096", "per_dispatch": {}}
for d in (f"pmc_fetch{csuf}", f"pmc_write{csuf}", f"pmc_sq{csuf}"):
    rows = list(csv.DictReader(open(glob.glob(str(g / f"{d}/runc/*counter_collection.csv"))[0]))); agg = {}
    for r in rows:
        if "celerite_scan" in r["Kernel_Name"]: agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in agg.items(): out["per_dispatch"][k] = sorted(v)[len(v) // 2]
pd = out["per_dispatch"]
out["derived"] = {"fetch_bytes_raw": pd["FETCH_SIZE"] * 1024, "fetch_bytes_gfx950_corrected_x2": pd["FETCH_SIZE"] * 2048,
                  "write_bytes": pd["WRITE_SIZE"] * 1024, "hbm_traffic_bytes": pd["FETCH_SIZE"] * 2048 + pd["WRITE_SIZE"] * 1024,
                  "valu_insts_per_wave_step": pd["SQ_INSTS_VALU"] / pd["SQ_WAVES"] / 9999,
                  "valu_busy_frac": pd["SQ_ACTIVE_INST_VALU"] * 4 / (pd["GRBM_GUI_ACTIVE"] / 8 * 1024),
                  "note": "FETCH_SIZE unit KB; x2 is the gfx950 correction of MI355X_MICROARCH.md (calibrated for 16 B/lane streams; "
                          "these are 8 B/lane buffer loads of an L2-resident 10 MB table, so the corrected figure is an upper bound). "
                          "valu_busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (GRBM_GUI_ACTIVE/8 cycles * 1024 SIMDs)."}
json.dump(out, open(prof / "r01_pmc_sho20_b4096.json", "w"), indent=1)
print(json.dumps(out["derived"]))
