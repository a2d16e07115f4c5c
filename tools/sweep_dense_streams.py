#!/usr/bin/env python3
"""GPU: pioran_dense_nll_batch with 1 .. 64 factorisations per batched launch (context option "dense_streams": until late round 3 the
number of concurrent single-matrix streams, now the matrices per launch, gridDim.z), N = 4096, J = 40; the single factorisation's value
is the cross-check."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
N, J = 4096, 40
t, y, yerr = bench.synth_series(10_000); t, y, yerr = t[:N], y[:N], yerr[:N]
ctx = pj.Context(0)
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (t[-1] - t[0]), 1 / (2 * np.min(np.diff(t))), J, 1.0, basis_function="SHO")
flop = N ** 3 / 3 + 2 * N ** 2
for Bd in (64,):
    A = np.tile(R.a, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]; Bb = np.tile(R.b, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]
    for ns in (1, 4, 8, 16, 32, 48, 64):
        ctx.set_option("dense_streams", ns)
        ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y))); ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        print(f"B={Bd} streams={ns}: {ms:.2f} ms per call, {ms / Bd:.3f} ms each, {Bd * flop / ms / 1e9:.1f} TFLOP/s = {Bd * flop / ms / 1e9 / 78.6:.3f} of the MFMA peak, finite={np.isfinite(v).all()}", flush=True)
    ctx.set_option("dense_streams", 1); v1 = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
    ctx.set_option("dense_streams", 32); v32 = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
    print(f"B={Bd}: batched launch vs one matrix per launch, max relative difference {np.max(np.abs(v32 - v1) / np.abs(v1)):.2e}", flush=True)

# pairs of steps (128-deep trailing updates) down to which size of the trailing matrix (in 64-row tiles), 32 matrices per launch
Bd = 64
A = np.tile(R.a, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]; Bb = np.tile(R.b, (Bd, 1)) * np.linspace(0.8, 1.2, Bd)[:, None]
ctx.set_option("dense_streams", 32)
for thr in (0, 2, 6, 12, 24, 48, 64):
    ctx.set_option("dense_batch_pair_threshold", thr)
    ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y))); ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    print(f"pairs while the trailing matrix has more than {thr} tiles per side: {ms / Bd:.3f} ms each", flush=True)
ctx.set_option("dense_batch_pair_threshold", -1)

# steps in fours (256-deep trailing updates) while the trailing matrix has more than thr tiles per side; batched and single
ref = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
for thr in (1 << 20, 48, 32, 24, 16, 8, 4, 0):
    ctx.set_option("dense_quad_threshold", thr)
    v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y)))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); v = ctx.dense_nll_batch(A, Bb, R.c, R.d, t, y, yerr ** 2, mu=np.full(Bd, np.mean(y))); ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    ts1 = []
    for _ in range(5):
        ts1.append(ctx.dense_nll_timed(R.a, R.b, R.c, R.d, t, y - np.mean(y), yerr ** 2)[2]["factor_ms"] if hasattr(ctx, "dense_nll_timed") else 0.0)
    print(f"fours above {thr} tiles per side: batched {ms / Bd:.3f} ms each (max rel diff {np.max(np.abs(v - ref) / np.abs(ref)):.1e}); one matrix: factorisation {np.median(ts1):.3f} ms", flush=True)
ctx.set_option("dense_quad_threshold", -1)
