#!/usr/bin/env python3
"""GPU: gradient, posterior mean (5000 times) and simulation with 1 .. 5 rows, four draws, N = 1e4: the step-by-step kernels (`no_block`; the
only path below six rows up to late round 4) against the windowed kernels (the automatic choice now).  ms per call incl. PCIe; deviations between the two."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
N = 10000
t, y, yerr = bench.synth_series(N); s2 = yerr ** 2
rng = np.random.default_rng(5)
def med(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for J, nreal in ((1, 1), (1, 0), (2, 0), (2, 2), (3, 1)):
    B = 4
    A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A; C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0, 3.0, J)
    if nreal: Bc[:, -nreal:] = 0; Dd[-nreal:] = 0
    mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2, B)
    ds = pj.Dataset(t, y, s2, ctx)
    res = {}
    tau = np.sort(rng.uniform(t[0], t[-1], 5000)); qn = rng.standard_normal((B, N))
    Cp = C[None, :] * rng.uniform(0.9, 1.1, (B, J)); Dp = Dd[None, :] * rng.uniform(0.9, 1.1, (B, J))
    for e in (0, 32):
        ctx.set_option("no_block", e == 0)        # 0: the step-by-step kernels (every call before late round 4), 32: the automatic choice
        g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu); kn = name()
        tg = med(lambda: ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu))
        pm = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu); kp = name()
        tp = med(lambda: ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu))
        sm = ctx.simulate(A, Bc, C, Dd, t, s2, qn); ks = name()
        ts = med(lambda: ctx.simulate(A, Bc, C, Dd, t, s2, qn))
        gp = ds.logl_grad(A, Bc, Cp, Dp, mu=mu, nu=nu); kgp = name()
        pp = ds.predict(A, Bc, Cp, Dp, tau, mu=mu, nu=nu); kpp = name()
        sp = ctx.simulate(A, Bc, Cp, Dp, t, s2, qn); ksp = name()
        res[e] = (g, kn, tg, pm, kp, tp, sm, ks, ts, gp, kgp, pp, kpp, sp, ksp)
    ctx.set_option("no_block", False)
    g0, g1 = res[0][0], res[32][0]
    dev = max(float(np.max(np.abs(np.asarray(g0[k], float) - np.asarray(g1[k], float))) / (1e-300 + np.max(np.abs(np.asarray(g0[k], float))))) for k in g0 if k.startswith("grad") or k == "logl")
    pdev = np.max(np.abs(res[0][3] - res[32][3])) / np.max(np.abs(res[0][3]))
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(a)))
    sdev = rel(res[0][6], res[32][6])
    gpdev = max(rel(np.asarray(res[0][9][k], float), np.asarray(res[32][9][k], float)) for k in res[0][9] if k.startswith("grad") or k == "logl")
    print(f"   simulate {res[0][8]:.2f} ms [{res[0][7]}] -> {res[32][8]:.2f} ms [{res[32][7]}] dev {sdev:.1e}; per-draw (c, d): gradient [{res[0][10]}] -> [{res[32][10]}] dev {gpdev:.1e}, "
          f"predict [{res[0][12]}] -> [{res[32][12]}] dev {rel(res[0][11], res[32][11]):.1e}, simulate [{res[0][14]}] -> [{res[32][14]}] dev {rel(res[0][13], res[32][13]):.1e}")
    print(f"J={J} nreal={nreal}: grad {res[0][2]:.2f} ms [{res[0][1]}] -> {res[32][2]:.2f} ms [{res[32][1]}] dev {dev}; predict {res[0][5]:.2f} ms [{res[0][4]}] -> {res[32][5]:.2f} ms [{res[32][4]}] dev {pdev:.1e}", flush=True)
    ds.close()
