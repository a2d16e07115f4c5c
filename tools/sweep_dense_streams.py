#!/usr/bin/env python3
"""GPU: pioran_dense_nll_batch with 4 .. 16 concurrent factorisations (context option "dense_streams"), N = 4096, J = 40."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
N, J = 4096, 40
t, y, yerr = bench.synth_series(10_000); t, y, yerr = t[:N], y[:N], yerr[:N]
ctx = pj.Context(0)
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (t[-1] - t[0]), 1 / (2 * np.min(np.diff(t))), J, 1.0, basis_function="SHO")
flop = N ** 3 / 3 + 2 * N ** 2
for Bd in (32, 64):
    A = np.tile(R.a, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]; Bb = np.tile(R.b, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]
    for ns in (4, 8, 12, 16):
        ctx.set_option("dense_streams", ns)
        ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y))); ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        print(f"B={Bd} streams={ns}: {ms:.2f} ms per call, {ms / Bd:.3f} ms each, {Bd * flop / ms / 1e9:.1f} TFLOP/s = {Bd * flop / ms / 1e9 / 78.6:.3f} of the MFMA peak, finite={np.isfinite(v).all()}", flush=True)
