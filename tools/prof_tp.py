#!/usr/bin/env python3
"""Time-parallel evaluations (celerite_tp.hip) for tools/kstats.sh: usage (GPU box): tools/kstats.sh tp tools/prof_tp.py <components> <segments, 0 = automatic> [basis = SHO] [draws = 1]
[scan = -1 automatic | 0 walk | 1 scan] — 20 forced calls at N = 1e4."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
N = 10000
nc = int(sys.argv[1]); segs = int(sys.argv[2])
basis = sys.argv[3] if len(sys.argv) > 3 else "SHO"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
scan = int(sys.argv[5]) if len(sys.argv) > 5 else -1
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(max(8, B), t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
ctx.set_option("scan_config", "tp"); ctx.set_option("tp_segments", segs); ctx.set_option("tp_scan", scan)
if os.environ.get("PIORAN_TP_SCAN_WAVES"): ctx.set_option("tp_scan_waves", int(os.environ["PIORAN_TP_SCAN_WAVES"]))     # (this tool's own switch: the library reads no such variable)
for _ in range(20): g = ds.logl_batch(A[:B], Bc[:B], C, Dd, mu=th[:B, 5].copy(), nu=th[:B, 4].copy())
print(pj._lib.lib().pioran_celerite_config_name(-1).decode())
