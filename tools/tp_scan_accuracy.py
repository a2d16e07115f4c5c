#!/usr/bin/env python3
"""GPU: accuracy of the time-parallel family with its boundary phase as a scan (option tp_scan = 1) against the sequential walk and the oracle over prior draws of the
bench models at N = 1e4, by conditioning ratio = nu min(sigma2) / sum(a) and segment count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
N = 10000
t, y, yerr = bench.synth_series(N)
nd = int(os.environ.get("DRAWS", 48))
th, f_min, f_max = bench.synth_theta(nd, t, y, seed=4321)
s2 = yerr ** 2
for basis, nc in (("SHO", 20), ("DRWCelerite", 10), ("SHO", 12), ("SHO", 8)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    mu, nu = th[:, 5].copy(), th[:, 4].copy()
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=16, return_status=True)
    ratio = nu * s2.min() / np.abs(A.sum(axis=1))
    ds = pj.Dataset(t, y, s2, ctx)
    res = {}
    for label, mode, segs in (("walk", 0, 0), ("scan32", 1, 32), ("scan64", 1, 64), ("scan128", 1, 128)):
        out = np.empty(nd)
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", mode); ctx.set_option("tp_segments", segs)
        for i in range(nd):
            out[i] = ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])[0]
        res[label] = np.abs(out - ref) / np.abs(ref)
    ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_segments", 0)
    ok = rst == 0
    print(f"# {basis}-{nc}: {ok.sum()} of {nd} draws positive definite; relative deviation from the fp64 oracle")
    edges = [0, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-2, 1e9]
    for lo, hi in zip(edges, edges[1:]):
        m = ok & (ratio >= lo) & (ratio < hi)
        if m.any():
            print(f"  ratio [{lo:.0e}, {hi:.0e}): {m.sum():3d} draws | " + " | ".join(f"{k} max {res[k][m].max():.1e} med {np.median(res[k][m]):.1e}" for k in res), flush=True)
