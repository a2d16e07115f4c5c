"""GPU: step-by-step vs two-step form of the throughput scan kernels, B = 4096, host-pointer entry (ms per call)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pioran_jl_amd as pj
from oracle import oracle as O

ctx = pj.Context(0)
N, B = 4000, 4096
t, y, yerr = O.synthetic_series(N)
rng = np.random.default_rng(3)
def timeit(f, n=3):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
name = lambda: pj._lib.lib().pioran_celerite_config_name(0).decode()
print("case rows config step_ms two_step_ms ratio maxrel")
cases = [("SHO", J, None) for J in (2, 4, 6, 7, 8, 10, 12, 15, 16, 20, 23, 24, 28, 31, 32, 39)] + [("DRW", 12, None), ("DRW", 20, None), ("real+", 20, 3)]
for kind, J, nreal in cases:
    if kind == "DRW":
        th = O.synthetic_theta(B, t, y)
        A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, J, "DRWCelerite")
    else:
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.005, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        if nreal:
            Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    res = {}
    for w2 in (False, True):
        ctx.set_option("win2", w2)
        res[w2] = (timeit(lambda: ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)), ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu), name())
    ctx.set_option("win2", False)
    ok = np.isfinite(res[False][1])
    rel = np.max(np.abs(res[True][1][ok] - res[False][1][ok]) / np.abs(res[False][1][ok]))
    print(f"{kind}-{J} {A.shape[1]} {res[True][2]} {res[False][0]:.3f} {res[True][0]:.3f} {res[False][0]/res[True][0]:.3f} {rel:.1e}", flush=True)
