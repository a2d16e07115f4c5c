#!/usr/bin/env python3
"""Gradient of log L at BASELINE size: N = 1e4, SHO-20 / DRWCelerite-20, B = 1 (one NUTS chain) and a
batch of chains; host-pointer entry (staging included), next to the value-only call and to one complex-step derivative of
the oracle (= the cost unit of one ForwardDiff partial on the CPU)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, J = 10_000, 20
basis = os.environ.get("BASIS", "SHO")
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(64, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
def timed(f, reps=3):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r
res = {"workload": f"N={N}, {basis}-{J}: value + gradient, host-pointer entry (PCIe included); 'abmunu': d/d(a_j, b_j, mu, nu); 'full': also "
                   "d/d(c_j, d_j); windowed reverse mode (default up to 63 rows) vs the step-by-step adjoint kernels (option no_block)"}
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
for B in (1, 16, 64, 256):
    tw, gw = timed(lambda: ds.logl_grad(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B], cd_grad=False)); kw = name()
    tf, gf = timed(lambda: ds.logl_grad(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B]))
    tv, v = timed(lambda: ds.logl_batch(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B]))
    row = {"windowed_abmunu_ms": round(tw, 2), "windowed_full_ms": round(tf, 2), "kernel": kw, "value_only_ms": round(tv, 2)}
    if B <= 64:
        ctx.set_option("no_block", True)
        tg, g = timed(lambda: ds.logl_grad(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B]))
        ctx.set_option("no_block", False)
        row["step_by_step_full_ms"] = round(tg, 2)
        row["max_rel_diff_windowed_vs_step_by_step"] = float(max(np.max(np.abs(gf[k] - g[k])) / (1 + np.max(np.abs(g[k])))
                                                               for k in ("grad_a", "grad_b", "grad_c", "grad_d", "grad_mu", "grad_nu")))
    res[f"B{B}"] = row
g = gw
# (c, d) per draw in every term (CARMA / QPO / free Celerite terms under NUTS): 16 chains
rngc = np.random.default_rng(5)
C2 = np.broadcast_to(C, (16, len(C))) * rngc.uniform(0.97, 1.03, (16, len(C))); D2 = np.broadcast_to(Dd, (16, len(C))) * rngc.uniform(0.97, 1.03, (16, len(C)))
tp, gp = timed(lambda: ds.logl_grad(A[:16], Bc[:16], C2, D2, mu=mu[:16], nu=nu[:16])); kp = name()
ctx.set_option("no_block", True)
tq, gq = timed(lambda: ds.logl_grad(A[:16], Bc[:16], C2, D2, mu=mu[:16], nu=nu[:16]), reps=1)
ctx.set_option("no_block", False)
okp = (gp["status"] == 0) & (gq["status"] == 0)
res["per_draw_cd_16_chains"] = {"windowed_full_ms": round(tp, 2), "kernel": kp, "step_by_step_draw_by_draw_ms": round(tq, 2),
                                "max_rel_diff": float(max(np.max(np.abs(gp[k][okp] - gq[k][okp])) / (1 + np.max(np.abs(gq[k][okp])))
                                                          for k in ("grad_a", "grad_b", "grad_c", "grad_d", "grad_mu", "grad_nu")))}
t0 = time.perf_counter(); ref = O.logl_dir(A[0], Bc[0], C, Dd, t, y - mu[0], nu[0] * yerr ** 2, da=np.ones(A.shape[1])); tc = time.perf_counter() - t0
res["cpu_one_complex_step_ms"] = round(tc * 1e3, 1)
res["directional_check_rel"] = float(abs(g["grad_a"][0].sum() - ref) / (1 + abs(ref)))
print(json.dumps(res))
