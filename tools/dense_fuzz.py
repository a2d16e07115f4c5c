#!/usr/bin/env python3
"""GPU: randomized comparison of the dense factorisation's schedules — one launch per block column (default), the panel / update chain of
rounds 1-3 (dense_old_chain = 1), single-panel / whole-tile variants and the persistent-chain prototype — on random sizes (129 .. 5200, any
remainder mod 64), term counts and amplitudes, plus matrices that stop being positive definite at a random pivot (same LAPACK-style info)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pioran_jl_amd as pj
from oracle import oracle as O

ctx = pj.Context(0)
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 120
worst = 0.0; t0 = time.time()
for idx in range(ncase):
    rng = np.random.default_rng([20261007, idx])
    N = int(rng.integers(129, 5200)) if idx % 4 else int(rng.choice([191, 192, 193, 255, 256, 257, 1471, 1472, 1473, 4096, 4097]))
    J = int(rng.integers(1, 12))
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    a = rng.uniform(0.1, 2, J); b = rng.uniform(-0.05, 0.05, J) * a; c = rng.uniform(0.05, 2, J); d = rng.uniform(0, 3, J)
    bad = int(rng.integers(0, N)) if idx % 5 == 0 else -1
    if bad >= 0: s2 = s2.copy(); s2[bad] = -50.0
    res = {}
    for name, opts in (("steps", {}), ("old", {"dense_old_chain": 1}), ("single", {"dense_no_pairs": True}), ("whole", {"dense_no_halves": True}), ("persistent", {"dense_old_chain": 5})):
        for k, v in opts.items(): ctx.set_option(k, v)
        res[name] = ctx.dense_nll(a, b, c, d, t, y, s2, return_info=True)
        for k in opts: ctx.set_option(k, 0 if k == "dense_old_chain" else False)
    v0, i0 = res["old"]
    for name, (v, i) in res.items():
        assert i == i0, (idx, N, J, bad, name, i, i0)
        if bad >= 0:
            assert i == bad + 1 and np.isnan(v), (idx, N, bad, name, v, i)
        else:
            dev = abs(v - v0) / abs(v0); worst = max(worst, dev)
            assert dev <= 1e-11, (idx, N, J, name, v, v0)
    if bad < 0 and N <= 260:   # (the oracle builds the covariance in Python: seconds per case beyond that)
        ref = O.dense_nll(a, b, c, d, t, y, s2)
        assert abs(res["steps"][0] - ref) <= 1e-10 * abs(ref), (idx, N, res["steps"][0], ref)
    if idx % 20 == 19: print(f"{idx + 1} cases, worst relative difference between schedules {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"{ncase} cases ok, worst relative difference between schedules {worst:.2e}")
