#!/usr/bin/env python3
"""GPU: d/d(c, d) of the many-chain reverse mode (celerite_tile.hip, round 6) against the small-batch windowed reverse mode: small shapes forced through the tile kernels,
then 4096 chains of the bench model with and without d/d(c, d), and 1024 chains on the small-batch kernels for the comparison."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(5)
for (J, N, B, nreal) in ((3, 50, 5, 0), (7, 100, 9, 2), (12, 333, 6, 0), (20, 200, 8, 0), (23, 97, 5, 3)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A; C = rng.uniform(0.05, 2, J); Dd = rng.uniform(0, 3, J)
    Bc[:, :nreal] = 0; Dd[:nreal] = 0
    mu = rng.normal(0, 0.1, B); nu = rng.uniform(0.5, 2, B)
    ds = pj.Dataset(t, y, s2, ctx)
    ctx.set_option("scan_config", "tile")
    g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
    k1 = name()
    ctx.set_option("scan_config", None)
    h = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
    k2 = name()
    out = [J, N, B, k1, '|', k2]
    for key in ("grad_a", "grad_b", "grad_c", "grad_d", "grad_mu", "grad_nu"):
        out.append(f"{key} {np.max(np.abs(g[key] - h[key])) / (1 + np.max(np.abs(h[key]))):.1e}")
    print(*out, flush=True)
# timing at the bench shape
import bench
t, y, yerr = bench.synth_series(10_000)
th, fmin, fmax = bench.synth_theta(4096, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], fmin, fmax, 20, th[:, 3], basis_function="SHO")
mu, nu = th[:, 5].copy(), th[:, 4].copy()
ds = pj.Dataset(t, y, yerr ** 2, ctx)
for cd in (False, True):
    ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd); k = name()
    w = []
    for _ in range(3):
        t0 = time.perf_counter(); g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd); w.append(time.perf_counter() - t0)
    print("4096 chains cd", cd, k, f"{min(w)*1e3:.1f} ms")
    if cd: gt = g
ctx.set_option("no_tile", True)
ds.logl_grad(A[:1024], Bc[:1024], C, Dd, mu=mu[:1024], nu=nu[:1024], cd_grad=True)
t0 = time.perf_counter(); h = ds.logl_grad(A[:1024], Bc[:1024], C, Dd, mu=mu[:1024], nu=nu[:1024], cd_grad=True); print("block 1024 chains cd", name(), f"{(time.perf_counter()-t0)*1e3:.1f} ms")
ok = (h["status"] == 0) & (gt["status"][:1024] == 0)
for key in ("grad_a", "grad_c", "grad_d"):
    sc = np.max(np.abs(h[key][ok]), axis=1) + 1
    d = np.max(np.abs(gt[key][:1024][ok] - h[key][ok]), axis=1) / sc
    print(key, "median", float(np.median(d)), "max", float(d.max()))
