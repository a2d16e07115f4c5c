#!/usr/bin/env python3
"""GPU: value + full gradient (a, b, c, d, mu, nu) at N = 1e4 by number of rows — which kernel family, ms per call (PCIe included)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
N = int(os.environ.get("N", 10000))
t, y, yerr = bench.synth_series(N)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
lib = pj._lib.lib()
for J in [int(x) for x in os.environ.get("JS", "20,31,32,40,47,48,56,64,71").split(",")]:
    for B in (1, 16):
        th = O.synthetic_theta(B, t, y, seed=J)
        A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, J, "SHO")
        g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
        kern = lib.pioran_celerite_config_name(-1).decode()
        v = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
        rel = np.max(np.abs(g["logl"] - v) / np.abs(v))
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps): ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
        ms = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps): ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
        msv = (time.perf_counter() - t0) / reps * 1e3
        print(f"J = {J:3d} rows = {2*J:3d} chains = {B:3d}: value + gradient {ms:8.2f} ms   value {msv:7.2f} ms   ratio {ms/msv:5.1f}   [{kern}]  value agrees {rel:.1e}", flush=True)
