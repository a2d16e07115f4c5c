#!/usr/bin/env python3
"""GPU: one larger sanity run of the paths added at the end of round 3 — prediction and simulation for 601 draws (three chunks), N = 5003
(not a multiple of 16 or 128), M = 2999 ascending and permuted evaluation times, against the oracle; a batched dense launch of 40
matrices at N = 3000 (steps in fours) against single calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, J, B, M = 5003, 20, 601, 2999
t, y, yerr = bench.synth_series(10_000); t, y, yerr = t[:N], y[:N], yerr[:N]
th, f_min, f_max = bench.synth_theta(B, t, y, seed=99)
f_min, f_max = 1 / (t[-1] - t[0]), 1 / (2 * np.min(np.diff(t)))
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
mu, nu = th[:, 5].copy(), th[:, 4].copy()
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
tau = np.sort(np.random.default_rng(1).uniform(t[0] - 5, t[-1] + 5, M))
t0 = time.perf_counter(); got, st = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu, return_status=True); print("predict", time.perf_counter() - t0, pj._lib.lib().pioran_celerite_config_name(-1).decode(), (st == 0).sum())
ok = np.flatnonzero(st == 0)
for i in (ok[0], ok[len(ok) // 2], ok[-1]):
    ref = O.predict(A[i], Bc[i], C, Dd, tau, t, y - mu[i], nu[i] * yerr ** 2) + mu[i]
    print(i, np.max(np.abs(got[i] - ref)) / np.max(np.abs(ref)))
q = np.random.default_rng(2).standard_normal((B, N))
t0 = time.perf_counter(); ys = ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); print("simulate", time.perf_counter() - t0, pj._lib.lib().pioran_celerite_config_name(-1).decode())
for i in (ok[0], ok[-1]):
    ref = O.sim(A[i], Bc[i], C, Dd, t, yerr ** 2, q[i]); print(i, np.max(np.abs(ys[i] - ref)) / np.max(np.abs(ref)))
# unsorted tau and M > N R fall back
perm = np.random.default_rng(3).permutation(M)
g2 = ds.predict(A[:5], Bc[:5], C, Dd, tau[perm], mu=mu[:5], nu=nu[:5])
print("unsorted vs sorted", np.nanmax(np.abs(g2 - got[:5][:, perm]) / (1e-300 + np.abs(got[:5][:, perm]))))
v = ctx.dense_nll_batch(A[:40] / 50, Bc[:40] / 50, C, Dd, t[:3000], y[:3000], yerr[:3000] ** 2 + 1.0, mu=mu[:40])
one = [ctx.dense_nll(A[i] / 50, Bc[i] / 50, C, Dd, t[:3000], y[:3000] - mu[i], yerr[:3000] ** 2 + 1.0) for i in (0, 39)]
print("dense batch vs single", abs(v[0] - one[0]) / abs(one[0]), abs(v[39] - one[1]) / abs(one[1]))
