#!/usr/bin/env python3
"""Build-container side of tools/run_profiles.sh: reads the rocprofv3 outputs merged back under gpurun_out/<round>/ and writes
the tracked summaries under profiles/ (<round>_*; round defaults to r03): kernel-trace stats of the bench command, the bench lines themselves, and
one PMC summary per workload carrying the kernel's full template signature, the configuration name and the fingerprint
of the kernel sources it was taken on (bench.py quotes `roofline.traffic` from it only while all three still match).
usage: python tools/collect_profiles.py [gpurun_out/r03 [r03]]"""
import csv, glob, json, shutil, sys
from pathlib import Path

root = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(root))
import bench  # noqa: E402  (scan_source_hash)

ROUND = sys.argv[2] if len(sys.argv) > 2 else "r06"
src = Path(sys.argv[1]) if len(sys.argv) > 1 else root / "gpurun_out" / ROUND
prof = root / "profiles"
for tag, basis in (("sho", "SHO"), ("drwcelerite", "DRWCelerite")):
    found = glob.glob(str(src / f"trace_{tag}" / "**" / "*kernel_stats.csv"), recursive=True)
    assert len(found) == 1, f"{len(found)} kernel-trace summaries under {src}/trace_{tag}: remove gpurun_out/<round> before a new tools/run_profiles.sh run"
    ks = found[0]
    shutil.copy(ks, prof / f"{ROUND}_bench_{tag}20_b4096_kernel_stats.csv")
    line = [ln for ln in (src / f"bench_plain_{tag}.json").read_text().splitlines() if ln.startswith("{")][-1]
    plain = json.loads(line)
    (prof / f"{ROUND}_bench_{tag}20_b4096.json").write_text(json.dumps(plain, indent=1))
    traced = json.loads([ln for ln in (src / f"bench_trace_{tag}.json").read_text().splitlines() if ln.startswith("{")][-1])
    rows = list(csv.DictReader(open(ks)))
    # the headline's main kernel: the step-by-step scan or (round 5) the windowed tile kernel — whichever carries the time; the tile kernel's
    # pre-pass (tile_pairs*_kernel) is reported beside it
    krow = max((r for r in rows if "celerite_scan_kernel" in r["Name"] or "celerite_tile_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
    pre = [r for r in rows if "tile_pairs" in r["Name"]]
    kmatch = "celerite_tile_kernel" if "celerite_tile_kernel" in krow["Name"] else "celerite_scan_kernel"
    out = {"command": f"rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --basis {basis} "
                      "--steps 2 --warmup 1   (one pass per counter group: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_*; tools/run_profiles.sh)",
           "kernel": krow["Name"], "kernel_config": plain["config"]["kernel_config"], "scan_source_hash": bench.scan_source_hash(),
           "workload": plain["config"]["workload"],
           "kernel_trace": {"calls": int(krow["Calls"]), "average_ms": float(krow["AverageNs"]) / 1e6, "min_ms": float(krow["MinNs"]) / 1e6,
                            "pre_pass": ({"kernel": pre[0]["Name"], "average_ms": float(pre[0]["AverageNs"]) / 1e6} if pre and kmatch == "celerite_tile_kernel" else None),
                            "bench_kernel_ms_same_run": traced["roofline"]["kernel_ms"], "bench_kernel_ms_unprofiled_run": plain["roofline"]["kernel_ms"],
                            "same_box_fp64_fma_ceiling_tflops": plain["roofline"].get("measured_fma_ceiling_tflops")},
           "per_dispatch": {}}
    for d in sorted(glob.glob(str(src / f"pmc_{tag}_*"))):
        if d.endswith(".err"):
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = {}
            for r in csv.DictReader(open(f)):
                if kmatch in r["Kernel_Name"]:
                    assert r["Kernel_Name"] == krow["Name"], (r["Kernel_Name"], krow["Name"])
                    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in agg.items():
                out["per_dispatch"][k] = sorted(v)[len(v) // 2]
    pd = out["per_dispatch"]
    N = plain["config"]["N"]
    out["derived"] = {
        "fetch_bytes_raw": pd["FETCH_SIZE"] * 1024, "fetch_bytes_gfx950_corrected_x2": pd["FETCH_SIZE"] * 2048,
        "write_bytes": pd["WRITE_SIZE"] * 1024, "hbm_traffic_bytes": pd["FETCH_SIZE"] * 2048 + pd["WRITE_SIZE"] * 1024,
        "valu_insts_per_wave_step": pd["SQ_INSTS_VALU"] / pd["SQ_WAVES"] / (N - 1),
        "fma_f64_share_of_valu": pd["SQ_INSTS_VALU_FMA_F64"] / pd["SQ_INSTS_VALU"], "mul_f64_share_of_valu": pd["SQ_INSTS_VALU_MUL_F64"] / pd["SQ_INSTS_VALU"],
        "flop_per_valu_lane_instruction": (2 * pd["SQ_INSTS_VALU_FMA_F64"] + pd["SQ_INSTS_VALU_MUL_F64"]) / pd["SQ_INSTS_VALU"],
        "valu_active_share_of_wave_cycles": pd["SQ_ACTIVE_INST_VALU"] / pd["SQ_WAVE_CYCLES"],
        "waitcnt_share_of_wave_cycles": pd.get("SQ_WAIT_ANY", float("nan")) / pd["SQ_WAVE_CYCLES"],
        "issue_wait_share_of_wave_cycles": pd.get("SQ_WAIT_INST_ANY", float("nan")) / pd["SQ_WAVE_CYCLES"],
        "mfma_f64_insts_per_wave_window": (pd["SQ_INSTS_MFMA"] / pd["SQ_WAVES"] / ((N + 15) // 16)) if "SQ_INSTS_MFMA" in pd else None,
        # share of the kernel's time in which a SIMD's matrix pipe is busy: (matrix instructions x 64 cycles each) / (1024 SIMDs x kernel cycles),
        # kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
        "mfma_pipe_busy_share": (pd["SQ_INSTS_MFMA"] * 64.0 / 1024.0 / (pd["GRBM_GUI_ACTIVE"] / 8.0)) if ("SQ_INSTS_MFMA" in pd and pd.get("GRBM_GUI_ACTIVE")) else None,
        "effective_clock_ghz": (pd["GRBM_GUI_ACTIVE"] / 8.0 / (float(krow["AverageNs"]))) if pd.get("GRBM_GUI_ACTIVE") else None,
        "note": "FETCH_SIZE / WRITE_SIZE in KB; x2 on the fetch is the gfx950 correction of MI355X_MICROARCH.md (calibrated on 16 B/lane streams; these "
                "are 8 B/lane buffer loads of an L2-resident table, so the corrected figure is an upper bound).  Median over the dispatches of the pass."}
    (prof / f"{ROUND}_pmc_{tag}20_b4096.json").write_text(json.dumps(out, indent=1))
    print(tag, json.dumps(out["derived"]), out["kernel_trace"])
