#!/usr/bin/env python3
"""GPU: the persistent-chain prototype of the dense factorisation (option dense_old_chain = 5: one resident workgroup runs the critical
role of every step, hand-offs through flags) against the launched chain: values at a range of sizes, event-timed factorisation at N = 4096."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O

ctx = pj.Context(0)
t, y, yerr = bench.synth_series(10_000)
rng = np.random.default_rng(5)
sizes = [int(x) for x in os.environ.get("SIZES", "192,193,256,320,449,1000,2049").split(",")]
for N in sizes:
    J = 6
    tt = np.cumsum(rng.uniform(0.05, 2.0, N)); yy = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    a = rng.uniform(0.1, 2, J); b = rng.uniform(-0.05, 0.05, J) * a; c = rng.uniform(0.05, 2, J); d = rng.uniform(0, 3, J)
    ctx.set_option("dense_old_chain", 0); new, i1 = ctx.dense_nll(a, b, c, d, tt, yy, s2, return_info=True)
    ctx.set_option("dense_old_chain", 5); per, i5 = ctx.dense_nll(a, b, c, d, tt, yy, s2, return_info=True)
    print({"N": N, "launched": new, "persistent": per, "info": [i1, i5], "rel": abs(new - per) / abs(new)}, flush=True)
N, J = 4096, 40
tt, yy, ee = t[:N], y[:N], yerr[:N]
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (tt[-1] - tt[0]), 1 / (2 * np.min(np.diff(tt))), J, 1.0, basis_function="SHO")
mu = float(np.mean(yy))
res = {}
for name, flag in (("launched", 0), ("persistent", 5), ("launched_again", 0), ("persistent_again", 5), ("chain_without_release_fence", 6),
                   ("chain_without_acquire_fence", 7), ("chain_without_either", 8)):
    ctx.set_option("dense_old_chain", flag)
    ctx.dense_nll(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)
    ph = [ctx.dense_nll_timed(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2) for _ in range(9)]
    res[name] = {"nll": ph[0][0], "info": ph[0][1], "factor_ms": float(np.median([p[2]["factor_ms"] for p in ph])),
                 "factor_ms_min": float(np.min([p[2]["factor_ms"] for p in ph]))}
    print(name, json.dumps(res[name]), flush=True)
ctx.set_option("dense_old_chain", 0)
print("rel persistent vs launched", abs(res["persistent"]["nll"] - res["launched"]["nll"]) / abs(res["launched"]["nll"]))
