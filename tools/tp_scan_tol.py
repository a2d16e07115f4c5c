#!/usr/bin/env python3
"""GPU: where to put the scan's acceptance threshold (option tp_scan_tol): per prior draw of the bench models at N = 1e4, B = 1, the error of the accepted result against the
oracle and whether the draw fell back to the walk (seen in the time of the call), for a ladder of thresholds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
N = 10000
t, y, yerr = bench.synth_series(N)
s2 = yerr ** 2
for seed, nd in ((4321, 8), (99, 40)):
    th, f_min, f_max = bench.synth_theta(nd, t, y, seed=seed) if nd != 8 else bench.synth_theta(8, t, y, seed=4321)
    for basis, nc in (("DRWCelerite", 10), ("SHO", 12), ("SHO", 20)):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
        mu, nu = th[:, 5].copy(), th[:, 4].copy()
        ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=16, return_status=True)
        ds = pj.Dataset(t, y, s2, ctx)
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 1)
        for tol in (1e-3, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13, 1e-14):
            ctx.set_option("tp_scan_tol", tol)
            errs, slow = [], 0
            for i in range(nd):
                if rst[i]: continue
                ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])
                t0 = time.perf_counter(); v = ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])[0]; dt = time.perf_counter() - t0
                errs.append(abs(v - ref[i]) / abs(ref[i])); slow += dt > 1.5e-3
            print(f"seed {seed} {basis}-{nc}: tol {tol:g}: max err {max(errs):.1e}, {slow} of {len(errs)} draws fell back to the walk", flush=True)
        ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_scan_tol", None)
