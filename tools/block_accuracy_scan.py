#!/usr/bin/env python3
"""GPU: deviation of the windowed kernel (and of the scan) from the oracle on PRIOR draws, against the draw's noise-to-signal ratio
nu * min(sigma2) / sum(a) — calibration of the host entries' re-evaluation threshold (capi.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
rows = []
for N, B, seed in ((150, 6000, 1), (1000, 3000, 2), (10000, 1024, 3)):
    if N == 10000:
        t, y, yerr = bench.synth_series(N)
    else:
        rng = np.random.default_rng(seed)
        t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); yerr = rng.uniform(0.01, 0.05, N)
    th = O.synthetic_theta(B, t, y, seed=seed)
    # widen the tail on purpose: a tenth of the draws with nu scaled down by 10 .. 1000
    rng = np.random.default_rng(seed + 10)
    k = rng.choice(B, B // 10, replace=False); th[k, 4] *= 10.0 ** (-rng.uniform(1, 3, len(k)))
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, "SHO")
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, yerr ** 2, mu, nu, nthreads=32, return_status=True)
    blk = np.empty(B); sc = np.empty(B); stb = np.empty(B, int)
    for b0 in range(0, B, 256):      # 256 at a time: the windowed kernel
        sl = slice(b0, min(B, b0 + 256))
        blk[sl], stb[sl] = ds.logl_batch(A[sl], Bc[sl], C, Dd, mu=mu[sl], nu=nu[sl], return_status=True)
    assert pj._lib.lib().pioran_celerite_config_name(-1).decode() == "block"
    ctx.set_option("no_block", True); ctx.set_option("no_wide", True)
    sc[:] = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
    ctx.set_option("no_block", False); ctx.set_option("no_wide", False)
    ok = (rst == 0) & (stb == 0) & np.isfinite(blk) & np.isfinite(sc)
    ratio = nu * np.min(yerr ** 2) / A.sum(axis=1)
    eb = np.abs(blk - ref) / np.abs(ref); es = np.abs(sc - ref) / np.abs(ref)
    print(f"N = {N}, {ok.sum()} positive-definite draws")
    edges = [0, 1e-8, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3, 1e-2, 1e9]
    for lo, hi in zip(edges, edges[1:]):
        m = ok & (ratio >= lo) & (ratio < hi)
        if m.any():
            print(f"  ratio [{lo:.0e}, {hi:.0e}): {m.sum():5d} draws   windowed max {eb[m].max():.1e} median {np.median(eb[m]):.1e} | scan max {es[m].max():.1e}")
    ds.close()
