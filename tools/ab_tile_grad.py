#!/usr/bin/env python3
"""celerite_tile.hip's reverse mode (one draw per wavefront) against the small-batch windowed reverse mode and the complex-step oracle, then
value + gradient at many chains, event-timed.  usage: python tools/ab_tile_grad.py [chains]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(5)
worst = 0.0
for J, N, B, nreal in ((3, 50, 5, 0), (8, 37, 6, 0), (20, 130, 7, 0), (12, 64, 4, 2), (20, 257, 9, 0), (28, 100, 5, 3), (31, 49, 3, 0), (10, 16, 4, 0), (10, 17, 4, 0)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.1, 3.0, J)
    if nreal: Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
    ds = pj.Dataset(t, y, s2, ctx)
    ctx.set_option("scan_config", "tile")
    g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=False); k1 = name()
    ctx.set_option("scan_config", None)
    h = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=False); k2 = name()
    errs = {}
    for key in ("logl", "grad_a", "grad_b", "grad_mu", "grad_nu"):
        sc = np.max(np.abs(h[key])) + 1e-300
        errs[key] = float(np.max(np.abs(g[key] - h[key])) / sc)
    og = O.logl_grad(A[0], Bc[0], C, Dd, t, y - mu[0], nu[0] * s2)
    ea = float(np.max(np.abs(g["grad_a"][0] - og["grad_a"])) / (np.max(np.abs(og["grad_a"])) + 1e-300))
    eb = float(np.max(np.abs(g["grad_b"][0] - og["grad_b"])) / (np.max(np.abs(og["grad_b"])) + 1e-300))
    worst = max(worst, max(errs.values()), ea, eb)
    print(f"J={J} N={N} B={B} nreal={nreal}: [{k1}] vs [{k2}]: " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()) + f" | vs complex step: a {ea:.1e} b {eb:.1e}", flush=True)
print("worst", worst)
if len(sys.argv) > 1:
    nch = int(sys.argv[1])
    N = 10000
    t, y, yerr = bench.synth_series(N)
    th, f_min, f_max = bench.synth_theta(nch, t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3], basis_function="SHO")
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    for cfg in ("tile", None):
        ctx.set_option("scan_config", cfg)
        if cfg is None: ctx.set_option("no_tile", True)
        g = ds.logl_grad(A, Bc, C, Dd, mu=th[:, 5].copy(), nu=th[:, 4].copy(), cd_grad=False)
        t0 = time.perf_counter()
        for _ in range(2): g = ds.logl_grad(A, Bc, C, Dd, mu=th[:, 5].copy(), nu=th[:, 4].copy(), cd_grad=False)
        ms = (time.perf_counter() - t0) / 2 * 1e3
        print(f"SHO-20 N={N} chains={nch} [{name()}]: {ms:.1f} ms per call incl. PCIe = {nch / ms:.1f} k value+gradient/s; finite {np.isfinite(g['grad_a']).all(axis=1).mean():.3f}", flush=True)
        if cfg == "tile": gt_ = g
        ctx.set_option("no_tile", False)
    ok = (g["status"] == 0) & (gt_["status"] == 0)
    for key in ("logl", "grad_a", "grad_b", "grad_mu", "grad_nu"):
        a_, b_ = gt_[key][ok], g[key][ok]
        sc = np.max(np.abs(b_).reshape(len(b_), -1), axis=1) + 1e-300
        d = np.max(np.abs(a_ - b_).reshape(len(b_), -1), axis=1) / sc
        print(f"  {key}: tile vs block over {ok.sum()} chains: max rel (per chain scale) {d.max():.2e}, median {np.median(d):.2e}")
