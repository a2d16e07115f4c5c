import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
N, J = 4096, 40
t, y, yerr = bench.synth_series(10_000); t, y, yerr = t[:N], y[:N], yerr[:N]
ctx = pj.Context(0)
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (t[-1] - t[0]), 1 / (2 * np.min(np.diff(t))), J, 1.0, basis_function="SHO")
Bd = 32
A = np.tile(R.a, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]; Bb = np.tile(R.b, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]
for _ in range(3):
    v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
print(np.isfinite(v).all())
