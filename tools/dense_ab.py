#!/usr/bin/env python3
"""GPU: the one-launch-per-column dense chain (dense_step_kernel) against the panel / update chain of rounds 1-3 (option
dense_old_chain) and the oracle: values at a range of sizes, event-timed factorisation at N = 4096 / 8192."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O

ctx = pj.Context(0)
t, y, yerr = bench.synth_series(10_000)
out = {"values": [], "timing": {}}
rng = np.random.default_rng(5)
for N in (130, 191, 192, 193, 256, 320, 449, 1000, 2049):
    J = 6
    tt = np.cumsum(rng.uniform(0.05, 2.0, N)); yy = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    a = rng.uniform(0.1, 2, J); b = rng.uniform(-0.05, 0.05, J) * a; c = rng.uniform(0.05, 2, J); d = rng.uniform(0, 3, J)
    ctx.set_option("dense_old_chain", False); new, i1 = ctx.dense_nll(a, b, c, d, tt, yy, s2, return_info=True)
    ctx.set_option("dense_old_chain", True); old, i0 = ctx.dense_nll(a, b, c, d, tt, yy, s2, return_info=True)
    ref = O.dense_nll(a, b, c, d, tt, yy, s2) if N <= 1000 else old
    out["values"].append({"N": N, "new": new, "old": old, "info": [i1, i0], "rel_new_vs_old": abs(new - old) / abs(old), "rel_new_vs_oracle": abs(new - ref) / abs(ref)})
    print(out["values"][-1], flush=True)
# non-PD: first bad pivot index must agree
tt = np.linspace(0, 100, 400); yy = np.ones(400); s2 = np.full(400, 1e-9)
for flag in (False, True):
    ctx.set_option("dense_old_chain", flag)
    print("non-PD", flag, ctx.dense_nll([1.0, -0.3], [0.0, 0.0], [0.3, 0.001], [0.0, 0.0], tt, yy, s2, return_info=True), flush=True)
for N in (4096, 8192):
    J = 40
    tt, yy, ee = t[:N], y[:N], yerr[:N]
    R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (tt[-1] - tt[0]), 1 / (2 * np.min(np.diff(tt))), J, 1.0, basis_function="SHO")
    mu = float(np.mean(yy))
    res = {}
    for name, flag in (("steps", False), ("old_chain", True), ("steps_again", False)):
        ctx.set_option("dense_old_chain", flag)
        ctx.dense_nll(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)
        ph = [ctx.dense_nll_timed(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2) for _ in range(9)]
        res[name] = {"nll": ph[0][0], "info": ph[0][1], "factor_ms": float(np.median([p[2]["factor_ms"] for p in ph])),
                     "build_ms": float(np.median([p[2]["build_ms"] for p in ph]))}
    flop = N ** 3 / 3 + 2 * N ** 2
    for k in res: res[k]["frac"] = flop / (res[k]["factor_ms"] * 1e-3) / 1e12 / 78.6
    res["rel_steps_vs_old"] = abs(res["steps"]["nll"] - res["old_chain"]["nll"]) / abs(res["old_chain"]["nll"])
    if N == 4096:
        cel = pj.log_likelihood(R, tt, yy - mu, ee ** 2, ctx=ctx)
        res["rel_vs_celerite"] = abs(cel + res["steps"]["nll"]) / abs(cel)
    out["timing"][str(N)] = res
    print(N, json.dumps(res), flush=True)
ctx.set_option("dense_old_chain", False)
print(json.dumps(out))
