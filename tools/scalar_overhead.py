import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
t, y, yerr = bench.synth_series(65536)
rng = np.random.Generator(np.random.PCG64(1234)); abcd = rng.random((64, 4)); abcd[:, 0] *= 5
def med(f, reps=41):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
for j in (2, 16):
    a, b, c, d = (np.ascontiguousarray(abcd[:j, k]) for k in range(4))
    for N in (32, 128, 512, 2048, 8192):
        tt, yy, ss = t[:N], y[:N], yerr[:N]
        r = []
        for e in (64, 0):
            ctx.set_option("exp", e)
            us = med(lambda: ctx.logl(a, b, c, d, tt, yy, ss)); v = ctx.logl(a, b, c, d, tt, yy, ss); r.append((us, v))
        ctx.set_option("exp", 0)
        ref = O.logl(a, b, c, d, tt, yy, ss)
        ds = pj.Dataset(tt, yy, ss, ctx); A = np.tile(a, (8, 1)); Bc = np.tile(b, (8, 1))
        rb = []
        for e in (64, 0):
            ctx.set_option("exp", e); rb.append(med(lambda: ds.logl_batch(A, Bc, c, d), 21))
        ctx.set_option("exp", 0); ds.close()
        print(f"j = {j:2d} N = {N:5d}: scalar call {r[0][0]:8.1f} us with copies -> {r[1][0]:8.1f} us reading / writing pinned host memory (same value: {r[0][1] == r[1][1]}, "
              f"rel. dev. from the oracle {abs(r[1][1] - ref) / abs(ref):.1e}); 8 draws on a resident series {rb[0]:8.1f} -> {rb[1]:8.1f} us", flush=True)
