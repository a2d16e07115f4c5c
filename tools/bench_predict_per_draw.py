#!/usr/bin/env python3
"""GPU: posterior mean and simulation for posterior draws with their OWN (c, d) in every term (QPO / CARMA / free Celerite models):
all draws in one launch of every kernel (per-draw windowed tables) against the draw-by-draw path (context option no_block: step-by-step
kernels, one call per draw).  N = 1e4, J = 20, M = 1e4."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, J = 10_000, 20; M = N
t, y, yerr = bench.synth_series(N)
out = {"workload": f"N={N}, SHO-{J} with (c, d) jittered per draw, M={M} evaluation times, host entries (transfers included)", "rows": []}
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
tau = np.linspace(t[0] - 10, t[-1] + 10, M)
for B in (4, 16, 64):
    th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
    rng = np.random.default_rng(B)
    C = np.tile(C, (B, 1)) * rng.uniform(0.9, 1.1, (B, J)); Dd = np.tile(Dd, (B, 1)) * rng.uniform(0.9, 1.1, (B, J))
    mu, nu = th[:, 5].copy(), th[:, 4].copy()
    q = rng.standard_normal((B, N))
    row = {"B": B}
    for label, nb in (("one_launch", "0"), ("draw_by_draw", "1")):
        ctx.set_option("no_block", nb)
        got = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); kp = pj._lib.lib().pioran_celerite_config_name(-1).decode()
        tp = []
        for _ in range(3 if label == "one_launch" else 1):
            t0 = time.perf_counter(); got = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); tp.append(time.perf_counter() - t0)
        tp = float(np.median(tp))
        ys = ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); ks = pj._lib.lib().pioran_celerite_config_name(-1).decode()
        tsim = []
        for _ in range(3 if label == "one_launch" else 1):
            t0 = time.perf_counter(); ys = ctx.simulate(A, Bc, C, Dd, t, yerr ** 2, q); tsim.append(time.perf_counter() - t0)
        tsim = float(np.median(tsim))
        row[label] = {"predict_ms": tp * 1e3, "predict_kernel": kp, "simulate_ms": tsim * 1e3, "simulate_kernel": ks}
        if label == "one_launch":
            _, st = ds.predict(A, Bc, C, Dd, tau[:8], mu=mu, nu=nu, return_status=True)
            i = int(np.flatnonzero(st == 0)[0])        # (prior draws: some are not positive definite — compare a valid one)
            ref = O.predict(A[i], Bc[i], C[i], Dd[i], tau, t, y - mu[i], nu[i] * yerr ** 2) + mu[i]
            row["predict_max_rel_err_vs_oracle"] = float(np.max(np.abs(got[i] - ref)) / np.max(np.abs(ref)))
            refs = O.sim(A[i], Bc[i], C[i], Dd[i], t, yerr ** 2, q[i])
            row["simulate_max_rel_err_vs_oracle"] = float(np.max(np.abs(ys[i] - refs)) / np.max(np.abs(refs)))
            row["valid_draws"] = int((st == 0).sum()); row["compared_draw"] = i
    ctx.set_option("no_block", "0")
    out["rows"].append(row)
    print(json.dumps(row), flush=True)
print(json.dumps(out))
