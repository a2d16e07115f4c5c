#!/usr/bin/env python3
"""GPU: randomized scalar calls (and a few draws through the batch entry) with 1 .. 5 positive definite terms, N = 1 .. 5000 and 16000 .. 20000,
against the oracle — the dispatch of late round 4 (windowed kernel for long series / the scalar call, coefficients read from pinned host memory).  100 s."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
worst = (0.0, None); kinds = {}
t0 = time.time(); n = 0
for idx in range(100000):
    if time.time() - t0 > 100: break
    rng = np.random.default_rng([777, idx])
    J = int(rng.integers(1, 6)); N = int(rng.integers(1, 5000)) if idx % 5 else int(rng.integers(16000, 20000))
    nreal = int(rng.integers(0, J + 1)) if rng.random() < 0.5 else 0
    t = np.cumsum(rng.uniform(0.01, 3.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(1e-3, 0.1, N)
    a = rng.uniform(0.05, 2.0, J); c = np.exp(rng.uniform(np.log(1e-3), np.log(20.0), J)); d = rng.uniform(0, 10.0, J)
    b = rng.uniform(-0.9, 0.9, J) * a * c / np.maximum(d, 1e-300)      # |b d| <= a c: a positive definite term
    b[:nreal] = 0; d[:nreal] = 0
    v, stv = ctx.logl(a, b, c, d, t, y, s2, return_status=True); k = name(); kinds[k] = kinds.get(k, 0) + 1
    r = O.logl(a, b, c, d, t, y, s2)
    if np.isfinite(r) and stv == 0:
        dev = abs(v - r) / max(1.0, abs(r))
        if dev > worst[0]: worst = (dev, dict(idx=idx, J=J, N=N, nreal=nreal, kernel=k, got=v, oracle=r))
    # a few draws on the same series through the batch entry (zero-copy coefficients)
    if idx % 3 == 0:
        B = int(rng.integers(2, 9)); A = rng.uniform(0.05, 2.0, (B, J)); Bc = rng.uniform(-0.9, 0.9, (B, J)) * A * c / np.maximum(d, 1e-300); Bc[:, :nreal] = 0
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        ds = pj.Dataset(t, y, s2, ctx); got, stb = ds.logl_batch(A, Bc, c, d, mu=mu, nu=nu, return_status=True); ds.close()
        ref = np.array([O.logl(A[i], Bc[i], c, d, t, y - mu[i], nu[i] * s2) for i in range(B)])
        ok = np.isfinite(ref) & (stb == 0)
        if ok.any():
            dev = float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))))
            if dev > worst[0]: worst = (dev, dict(idx=idx, J=J, N=N, nreal=nreal, kernel=name(), batch=B))
    n += 1
print("cases", n, "kernels of the scalar calls", kinds, "worst relative deviation", worst)
assert worst[0] < 1e-8
