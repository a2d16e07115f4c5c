#!/usr/bin/env python3
"""GPU: the time-parallel family on the ill-conditioned draws of tests/golden/quad_truth.npz (the __float128 truth): boundary phase as the sequential walk against
the scan (option tp_scan = 1, automatic segment counts), by bin of ratio = nu min(sigma2) / sum(a)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import pioran_jl_amd as pj
ctx = pj.Context(0)
q = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "quad_truth.npz"))
for N in (150, 1000):
    tag = f"n{N}"
    t, y, yerr = q[f"{tag}_t"], q[f"{tag}_y"], q[f"{tag}_yerr"]
    A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
    truth, ratio, orc = q[f"{tag}_truth"], q[f"{tag}_ratio"], q[f"{tag}_oracle_fp64"]
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    res = {}
    for label, mode in (("walk", 0), ("scan", 1)):
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", mode)
        out = np.empty(len(truth))
        for i in range(len(truth)):
            out[i] = ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])[0]
        res[label] = np.abs(out - truth) / np.abs(truth)
    ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1)
    eo = np.abs(orc - truth) / np.abs(truth)
    print(f"# N = {N}: {len(truth)} draws, relative deviation from the __float128 truth (max / median)")
    edges = [0, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1]
    for lo, hi in zip(edges, edges[1:]):
        m = (ratio >= lo) & (ratio < hi)
        if m.any():
            print(f"  ratio [{lo:.0e}, {hi:.0e}): {m.sum():3d} draws | fp64 oracle {eo[m].max():.1e} / {np.median(eo[m]):.1e} | " + " | ".join(f"tp {k} {res[k][m].max():.1e} / {np.median(res[k][m]):.1e}" for k in res), flush=True)
