#!/usr/bin/env python3
"""GPU: value + gradient for many chains (SHO-20, N = 1e4) — ms per call by number of chains, with and without d/d(c, d)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
t, y, yerr = bench.synth_series(10000)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
for B in [int(x) for x in os.environ.get("BS", "64,256,512,1024,4096").split(",")]:
    th = O.synthetic_theta(B, t, y, seed=5)
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, "SHO")
    for cd in (True, False):
        g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps): ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd)
        ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"chains = {B:5d} cd = {cd!s:5s}: {ms:8.2f} ms  {B / ms:7.2f} k value+gradients/s  finite {np.isfinite(g['grad_a']).mean():.3f}", flush=True)
