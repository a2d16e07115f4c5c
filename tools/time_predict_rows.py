#!/usr/bin/env python3
"""GPU: posterior mean at M = N = 1e4 times and simulation for 16 draws by number of rows — kernel family and ms per call (PCIe included)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
N = 10000
t, y, yerr = bench.synth_series(N)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
lib = pj._lib.lib()
tau = np.linspace(t[0] - 10, t[-1] + 10, N)
B = int(os.environ.get("B", 16))
for J in [int(x) for x in os.environ.get("JS", "20,31,32,40,47,48,64,71").split(",")]:
    th = O.synthetic_theta(B, t, y, seed=J)
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, J, "SHO")
    q = np.random.default_rng(J).standard_normal((B, N))
    m = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); kp = lib.pioran_celerite_config_name(-1).decode()
    t0 = time.perf_counter(); ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); tp = (time.perf_counter() - t0) / 2 * 1e3
    ys = ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); ks = lib.pioran_celerite_config_name(-1).decode()
    t0 = time.perf_counter(); ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); ts = (time.perf_counter() - t0) / 2 * 1e3
    print(f"J = {J:3d} rows = {2*J:3d} draws = {B}: posterior mean {tp:8.2f} ms [{kp}]   simulation {ts:8.2f} ms [{ks}]   finite {np.isfinite(m).all() and np.isfinite(ys).all()}", flush=True)
