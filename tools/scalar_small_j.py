#!/usr/bin/env python3
"""GPU: the scalar drop-in `logl` with fewer than six rows (1 .. 2 terms; reference grid j = 2, benchmark/benchmarks.jl:16-18):
the automatic choice against the windowed kernel forced by name, N = 2^10 .. 2^16.  ms per call (PCIe included), median of 7."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
from oracle import oracle as O

t, y, yerr = bench.synth_series(65536)
rng = np.random.Generator(np.random.PCG64(1234))
abcd = rng.random((64, 4)); abcd[:, 0] *= 5
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()


def med_ms(f, reps=7):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for j, real in ((1, False), (1, True), (2, False), (2, True), (3, True)):
    a, b, c, d = (np.ascontiguousarray(abcd[:j, k]) for k in range(4))
    if real:
        b = np.zeros(j); d = np.zeros(j)
    for N in (1024, 8192, 16384, 32768, 65536):
        tt, yy, ss = t[:N], y[:N], yerr[:N]
        out = []
        for cfg in (None, "block"):
            ctx.set_option("scan_config", cfg)
            ms = med_ms(lambda: ctx.logl(a, b, c, d, tt, yy, ss)); v = ctx.logl(a, b, c, d, tt, yy, ss); out.append((ms, v, name()))
        ctx.set_option("scan_config", None)
        ref = O.logl(a, b, c, d, tt, yy, ss)
        print(f"j = {j} one-row terms = {real!s:5s} N = {N:6d}: automatic {out[0][0]:7.3f} ms [{out[0][2]}] | windowed {out[1][0]:7.3f} ms [{out[1][2]}] | "
              f"rel. dev. from the oracle {abs(out[0][1] - ref) / abs(ref):.1e} / {abs(out[1][1] - ref) / abs(ref):.1e}", flush=True)

# batches sharing (c, d): where the windowed kernel stops winning below six rows
for N in (10000, 65536):
    tt, yy, ss = t[:N], y[:N], yerr[:N]
    ds = pj.Dataset(tt, yy, ss, ctx)
    for j, real in ((1, True), (2, False), (5, True)):
        c, d = np.ascontiguousarray(abcd[:j, 2]), (np.zeros(j) if real else np.ascontiguousarray(abcd[:j, 3]))
        for B in (1, 256, 512, 1024):
            A = rng.random((B, j)) * 5; Bc = np.zeros((B, j)) if real else rng.random((B, j))
            res = []
            for cfg in (None, "block"):
                ctx.set_option("scan_config", cfg)
                res.append((med_ms(lambda: ds.logl_batch(A, Bc, c, d), 5), name()))
            ctx.set_option("scan_config", None)
            print(f"N = {N} rows = {j if real else 2 * j} draws = {B:5d}: automatic {res[0][0]:7.3f} ms [{res[0][1]}] | windowed {res[1][0]:7.3f} ms [{res[1][1]}]", flush=True)
    ds.close()
