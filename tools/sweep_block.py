"""Scratch (GPU): where does the windowed kernel win?  ms per call, device-resident launches excluded: host-pointer entry."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pioran_jl_amd as pj
from oracle import oracle as O

ctx = pj.Context(0)
N = 10_000
t, y, yerr = O.synthetic_series(N)
rng = np.random.default_rng(3)
def timeit(f, n=4):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("J R B block_ms other_ms (other = latency layout for B <= 256 and R >= 16, else throughput layouts)")
for J in (2, 4, 7, 8, 12, 16, 20, 24, 30, 31):
    for B in (1, 64, 256, 512, 1024):
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.005, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        ds = pj.Dataset(t, y, yerr ** 2, ctx)
        ctx.set_option("scan_config", "block")
        tb = timeit(lambda: ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu))
        g1 = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
        ctx.set_option("scan_config", None); ctx.set_option("no_block", True)
        to = timeit(lambda: ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu))
        g2 = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
        ctx.set_option("no_block", False)
        print(f"{J:2d} {2*J:2d} {B:5d} {tb:8.3f} {to:8.3f}  ratio {to/tb:5.2f}  maxrel {np.max(np.abs(g1-g2)/np.abs(g2)):.1e}", flush=True)
