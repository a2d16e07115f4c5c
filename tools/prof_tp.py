#!/usr/bin/env python3
"""One forced time-parallel evaluation (celerite_tp.hip) for tools/kstats.sh: usage (GPU box): tools/kstats.sh tp tools/prof_tp.py <SHO components> <segments>"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
N = 10000
nc = int(sys.argv[1]); segs = int(sys.argv[2])
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(8, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function="SHO")
ds = pj.Dataset(t, y, yerr ** 2, ctx)
ctx.set_option("scan_config", "tp"); ctx.set_option("tp_segments", segs)
for _ in range(3): g = ds.logl_batch(A[:1], Bc[:1], C, Dd, mu=th[:1, 5].copy(), nu=th[:1, 4].copy())
print(pj._lib.lib().pioran_celerite_config_name(-1).decode())
