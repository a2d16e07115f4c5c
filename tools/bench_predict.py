#!/usr/bin/env python3
"""Posterior-mean prediction and simulation at BASELINE size: B posterior draws, N = 1e4 data points,
M evaluation times, SHO-20; host-pointer entries (PCIe and host staging included), next to the oracle on one core."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, J = 10_000, 20
B = int(os.environ.get("B", "256")); M = int(os.environ.get("M", "10000"))
basis = os.environ.get("BASIS", "SHO")
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
tau = np.linspace(t[0] - 10, t[-1] + 10, M)
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
def timed(f, reps=3):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r
tp, got = timed(lambda: ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu))
q = np.random.default_rng(1).standard_normal((B, N))
tsim, ys = timed(lambda: ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q))
t0 = time.perf_counter(); ref = O.predict(A[0], Bc[0], C, Dd, tau, t, y - mu[0], nu[0] * yerr ** 2) + mu[0]; tcp = time.perf_counter() - t0
t0 = time.perf_counter(); rs = O.sim(A[0], Bc[0], C, Dd, t, yerr ** 2, q[0]); tcs = time.perf_counter() - t0
print(json.dumps({"workload": f"N={N}, {basis}-{J}, B={B} draws, M={M} evaluation times",
                  "predict_ms_per_call": tp * 1e3, "predict_draws_per_s": B / tp, "predict_cpu_one_core_ms_per_draw": tcp * 1e3,
                  "predict_max_rel_err_vs_oracle": float(np.max(np.abs(got[0] - ref)) / np.max(np.abs(ref))),
                  "simulate_ms_per_call": tsim * 1e3, "simulate_draws_per_s": B / tsim, "simulate_cpu_one_core_ms_per_draw": tcs * 1e3,
                  "simulate_max_rel_err_vs_oracle": float(np.max(np.abs(ys[0] - rs)) / np.max(np.abs(rs)))}))
