#!/usr/bin/env python3
"""GPU: every kernel family and the fp64 oracle against the __float128 truth on the ill-conditioned draws of tests/golden/quad_truth.npz
(oracle/make_quad_truth.py), per bin of ratio = nu min(sigma2) / sum(a) ~ 1 / cond(K).  Settles whose error the deviations between the
kernels and the fp64 oracle in those bins are (profiles/r05_quad_truth.txt)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import pioran_jl_amd as pj
from oracle import oracle as O
q = np.load("tests/golden/quad_truth.npz")
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
print("# relative deviation from the __float128 evaluation (oracle/celerite_oracle_q.c), max / median per bin; SHO-20 prior draws, a tenth with nu scaled down")
for N in (150, 1000, 10000):
    tag = f"n{N}"
    if N == 10000:
        t, y, yerr = O.synthetic_series(N, seed=1234)
    else:
        t, y, yerr = q[f"{tag}_t"], q[f"{tag}_y"], q[f"{tag}_yerr"]
    A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
    truth, ratio, orc = q[f"{tag}_truth"], q[f"{tag}_ratio"], q[f"{tag}_oracle_fp64"]
    B = len(truth)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    res = {"fp64 oracle": orc}
    ctx.set_option("no_block", True); ctx.set_option("no_wide", True)
    res["scan"] = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu); assert name() == "scan"
    ctx.set_option("no_block", False); ctx.set_option("no_wide", False)
    ctx.set_option("scan_config", "block")
    res["windowed (block)"], stb = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True); assert name() == "block"
    ctx.set_option("scan_config", "tile")
    res["windowed (tile)"], stt = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True); assert name() == "tile"
    ctx.set_option("scan_config", "tp")         # the time-parallel family (celerite_tp.hip), 64 draws per call
    tpv = np.empty(B); stp = np.empty(B, dtype=np.int32)
    for b0 in range(0, B, 64):
        sl = slice(b0, min(B, b0 + 64))
        tpv[sl], stp[sl] = ds.logl_batch(A[sl], Bc[sl], C, Dd, mu=mu[sl], nu=nu[sl], return_status=True); assert name() == "tp"
    res["time-parallel"] = tpv
    ctx.set_option("scan_config", None)
    print(f"N = {N}: {B} draws; status: block {int((stb != 0).sum())} flagged, tile {int((stt != 0).sum())} flagged, time-parallel {int((stp != 0).sum())} flagged")
    edges = [0, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5]
    for lo, hi in zip(edges, edges[1:]):
        m = (ratio >= lo) & (ratio < hi)
        if not m.any():
            continue
        line = f"  ratio [{lo:.0e}, {hi:.0e}): {m.sum():4d} draws |"
        for k, v in res.items():
            e = np.abs(v[m] - truth[m]) / np.abs(truth[m])
            line += f" {k} {np.nanmax(e):.1e} / {np.nanmedian(e):.1e} |"
        print(line)
    ds.close()
