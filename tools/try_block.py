"""Scratch driver (GPU): block kernel vs oracle and vs the latency layout."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pioran_jl_amd as pj
from oracle import oracle as O

def relerr(a, b):
    return np.max(np.abs(a - b) / np.maximum(1e-300, np.abs(b)))

ctx = pj.Context(0)
rng = np.random.default_rng(5)
for J in [1, 2, 3, 5, 7, 8, 10, 13, 16, 20, 21, 24, 27, 32, 39]:
    for N in [1, 5, 16, 17, 257]:
        B = 9
        t = np.cumsum(rng.uniform(0.01, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        ds = pj.Dataset(t, y, s2, ctx)
        ctx.set_option("scan_config", "block")
        got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True)
        ctx.set_option("scan_config", None)
        ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=4)
        e = relerr(got, ref)
        flag = "" if e < 1e-10 else "  <-- BAD"
        print(f"J={J:2d} N={N:4d} rel {e:.2e} status {st.max()}{flag}", flush=True)

# benchmark series
t, y, yerr = O.synthetic_series(10_000)
for basis in ("SHO", "DRWCelerite"):
    th = O.synthetic_theta(256, t, y)
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, basis)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    ref = O.logl_batch(A[:32], Bc[:32], C, Dd, t, y, yerr ** 2, mu[:32], nu[:32], nthreads=8)
    for cfg in ("block", "wide"):
        ctx.set_option("scan_config", cfg)
        for B in (1, 32, 256):
            got = ds.logl_batch(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B])
            t0 = time.perf_counter()
            for _ in range(5):
                got = ds.logl_batch(A[:B], Bc[:B], C, Dd, mu=mu[:B], nu=nu[:B])
            dt = (time.perf_counter() - t0) / 5
            nb = min(B, 32)
            print(f"{basis} {cfg} B={B}: {dt*1e3:.3f} ms per call, rel err {relerr(got[:nb], ref[:nb]):.2e}", flush=True)
    ctx.set_option("scan_config", None)
