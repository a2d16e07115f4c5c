"""Value + gradient of 4096 chains through the host entry, best of 5 (ms): SHO-20 (three block columns) and, with `drw`, DRWCelerite-20 (four);
`cd`: with d/d(c, d).  For tools/ab_variant_run_grad.sh (same-box A/B of a compile-time switch through PIORAN_HIP_LIB)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
N = 10000; nch = 4096
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(nch, t, y, seed=4321)
basis = "DRWCelerite" if "drw" in sys.argv[1:] else "SHO"
cd = "cd" in sys.argv[1:]
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3], basis_function=basis)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
best = 1e9
for _ in range(6):
    t0 = time.perf_counter()
    g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, cd_grad=cd)
    best = min(best, time.perf_counter() - t0)
print(f"{best * 1e3:.2f}")
