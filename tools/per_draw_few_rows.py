#!/usr/bin/env python3
"""GPU: (c, d) per draw in every term with fewer than six rows — the automatic choice against the windowed kernel forced by name
(one or two terms: per-draw rows; more: one table per draw).  ms per call incl. PCIe."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(3)
def med(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for N in (1000, 10000, 65536):
    t, y, yerr = bench.synth_series(N); s2 = yerr ** 2
    ds = pj.Dataset(t, y, s2, ctx)
    for J, real in ((1, False), (2, False), (3, True), (5, True)):
        for B in (1, 16, 64, 256, 768):
            A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A; C = rng.uniform(0.05, 2.0, (B, J)); Dd = rng.uniform(0, 3.0, (B, J))
            if real:      # one-row terms only (Exp / DRW with free time scales): J rows
                Bc[:] = 0.0; Dd[:] = 0.0
            r = []
            for cfg in (None, "block"):
                ctx.set_option("scan_config", cfg)
                v = ds.logl_batch(A, Bc, C, Dd); r.append((med(lambda: ds.logl_batch(A, Bc, C, Dd)), name(), v))
            ctx.set_option("scan_config", None)
            print(f"N = {N} J = {J}{' (one-row terms)' if real else ''} draws = {B}: automatic {r[0][0]:.3f} ms [{r[0][1]}] | windowed {r[1][0]:.3f} ms [{r[1][1]}] | max rel diff {np.max(np.abs(r[0][2]-r[1][2])/np.abs(r[0][2])):.1e}", flush=True)
    ds.close()
