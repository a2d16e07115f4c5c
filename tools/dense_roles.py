#!/usr/bin/env python3
"""GPU: which role bounds a step of dense_step_kernel — the factorisation timed with a subset of the roles launched (results are garbage;
option dense_old_chain = 2 critical workgroup alone, 3 DIAG2 + strips alone, 4 bulk alone)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
t, y, yerr = bench.synth_series(10_000)
N, J = 4096, 40
tt, yy, ee = t[:N], y[:N], yerr[:N]
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (tt[-1] - tt[0]), 1 / (2 * np.min(np.diff(tt))), J, 1.0, basis_function="SHO")
mu = float(np.mean(yy))
for name, mode in (("all roles", 0), ("critical workgroup alone", 2), ("DIAG2 + strips alone", 3), ("bulk alone", 4), ("all roles", 0)):
    ctx.set_option("dense_old_chain", mode)
    ctx.dense_nll(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)
    ph = [ctx.dense_nll_timed(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)[2]["factor_ms"] for _ in range(7)]
    print(f"{name:28s}: factorisation {np.median(ph):.3f} ms = {np.median(ph) * 1e3 / 64:.1f} us per step", flush=True)
ctx.set_option("dense_old_chain", 0)
