#!/usr/bin/env python3
"""GPU: the many-chain reverse mode at FOUR block columns (48 .. 63 state rows, three draws per workgroup: celerite_tile_adjoint_kernel<4>) against the small-batch
windowed reverse mode: small shapes forced through the tile kernels, then 4096 chains of DRWCelerite-20 (60 rows) at N = 1e4 with and without d/d(c, d)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import pioran_jl_amd as pj
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(5)
for (J, N, B, nreal) in ((24, 50, 5, 0), (28, 100, 9, 2), (31, 333, 7, 0), (40, 200, 8, 20), (30, 97, 5, 3), (25, 64, 4, 1)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A; C = rng.uniform(0.05, 2, J); Dd = rng.uniform(0, 3, J)
    Bc[:, :nreal] = 0; Dd[:nreal] = 0
    mu = rng.normal(0, 0.1, B); nu = rng.uniform(0.5, 2, B)
    ds = pj.Dataset(t, y, s2, ctx)
    for cd in (True, False):
        ctx.set_option("scan_config", "tile")
        g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd)
        k1 = name()
        ctx.set_option("scan_config", None)
        h = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd)
        k2 = name()
        out = [2 * J - nreal, N, B, k1, '|', k2, f"logl {np.max(np.abs(g['logl'] - h['logl']) / np.abs(h['logl'])):.1e}"]
        for key in ("grad_a", "grad_b", "grad_c", "grad_d", "grad_mu", "grad_nu") if cd else ("grad_a", "grad_b", "grad_mu", "grad_nu"):
            out.append(f"{key} {np.max(np.abs(g[key] - h[key])) / (1 + np.max(np.abs(h[key]))):.1e}")
        print(*out, flush=True)
import bench
t, y, yerr = bench.synth_series(10_000)
th, fmin, fmax = bench.synth_theta(4096, t, y, seed=4321)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
ds = pj.Dataset(t, y, yerr ** 2, ctx)
for basis, nc in (("DRWCelerite", 20), ("SHO", 28), ("SHO", 20)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], fmin, fmax, nc, th[:, 3], basis_function=basis)
    ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
    w = []
    for _ in range(3):
        t0 = time.perf_counter(); ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu); w.append(time.perf_counter() - t0)
    val = min(w) * 1e3
    print(f"{basis}-{nc}: 4096 values [{name()}] {val:.1f} ms (host entry)", flush=True)
    for cd in (False, True):
        res = {}
        for label, opt in (("tile", False), ("small-batch", True)):
            ctx.set_option("no_tile", opt); ctx.set_option("scan_config", None if opt else "tile")      # (the tile reverse mode forced: not the automatic choice past 47 rows)
            nb = 4096 if not opt else 1024
            g = ds.logl_grad(A[:nb], Bc[:nb], C, Dd, mu=mu[:nb], nu=nu[:nb], cd_grad=cd); k = name()
            w = []
            for _ in range(2):
                t0 = time.perf_counter(); g = ds.logl_grad(A[:nb], Bc[:nb], C, Dd, mu=mu[:nb], nu=nu[:nb], cd_grad=cd); w.append(time.perf_counter() - t0)
            res[label] = g
            print(f"{basis}-{nc}: {nb} chains d/d(c,d)={cd} [{k}] {min(w) * 1e3:.1f} ms" + (f" = {min(w) * 1e3 / val:.2f} x the values" if not opt else f" (x 4 = {4 * min(w) * 1e3:.1f})"), flush=True)
        ctx.set_option("no_tile", False); ctx.set_option("scan_config", None)
        gt, h = res["tile"], res["small-batch"]
        ok = (h["status"] == 0) & (gt["status"][:1024] == 0)
        for key in ("grad_a", "grad_c", "grad_d") if cd else ("grad_a",):
            sc = np.max(np.abs(h[key][ok]), axis=1) + 1
            d = np.max(np.abs(gt[key][:1024][ok] - h[key][ok]), axis=1) / sc
            print("   ", key, "tile vs small-batch over 1024 chains: median", float(np.median(d)), "max", float(d.max()), flush=True)
