#!/usr/bin/env python3
"""celerite_tp.hip (time-parallel evaluation of a handful of draws) against the oracle and the serial-chain kernels: accuracy, then time per call.
usage: python tools/ab_tp.py [time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(11)
worst = 0.0
for J, N, B, nreal, nseg in ((2, 200, 1, 0, 4), (3, 500, 2, 1, 5), (6, 300, 3, 0, 0), (20, 1000, 2, 0, 8), (20, 777, 1, 0, 3), (9, 400, 4, 4, 6), (24, 640, 2, 0, 4),
                             (21, 500, 1, 20, 5), (1, 100, 1, 0, 2), (1, 100, 1, 1, 2), (12, 2000, 8, 3, 0)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.1, 3.0, J)
    if nreal: Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
    ds = pj.Dataset(t, y, s2, ctx)
    ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=4)
    ctx.set_option("scan_config", "tp"); ctx.set_option("tp_segments", nseg)
    got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True); k = name()
    ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0)
    e = float(np.max(np.abs(got - ref) / np.abs(ref)))
    worst = max(worst, e)
    print(f"J={J} N={N} B={B} nreal={nreal} nseg={nseg}: [{k}] max rel vs oracle {e:.2e} status {st.tolist()}", flush=True)
print("worst", worst)
if len(sys.argv) > 1:
    N = 10000
    t, y, yerr = bench.synth_series(N)
    th, f_min, f_max = bench.synth_theta(8, t, y, seed=4321)
    for basis, nc in (("SHO", 20), ("DRWCelerite", 10), ("SHO", 2), ("SHO", 8)):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
        ds = pj.Dataset(t, y, yerr ** 2, ctx)
        for nb in (1, 8):
            ref = O.logl_batch(A[:nb], Bc[:nb], C, Dd, t, y, yerr ** 2, th[:nb, 5].copy(), th[:nb, 4].copy(), nthreads=4)
            for cfg, segs in ((None, [0]), ("tp", [0, 8, 16, 24, 32, 48, 64])):
                for sg in segs:
                    ctx.set_option("scan_config", cfg); ctx.set_option("tp_segments", sg)
                    try:
                        got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy())
                    except Exception as ex:
                        print(f"{basis}-{nc} B={nb} {cfg} segs={sg}: {ex}"); continue
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter(); got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy()); ts.append(time.perf_counter() - t0)
                    e = float(np.max(np.abs(got - ref) / np.abs(ref)))
                    print(f"{basis}-{nc} N={N} B={nb} [{name()}] segs={sg}: {min(ts) * 1e3:.3f} ms per call (host entry, PCIe included); max rel vs oracle {e:.2e}", flush=True)
            ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0)
if len(sys.argv) > 2:      # sweep: where does the time-parallel family win?  (rows <= 16)
    for N in (512, 1024, 2048, 4096, 8192, 10000):
        t, y, yerr = bench.synth_series(N)
        th, f_min, f_max = bench.synth_theta(64, t, y, seed=4321)
        for nc in (1, 2, 4, 6, 8):
            A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function="SHO")
            ds = pj.Dataset(t, y, yerr ** 2, ctx)
            line = f"N={N} SHO-{nc}:"
            for nb in (1, 8, 64):
                res = {}
                for cfg in (None, "tp"):
                    ctx.set_option("scan_config", cfg)
                    got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy())
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter(); got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy()); ts.append(time.perf_counter() - t0)
                    res[cfg] = (min(ts) * 1e3, name())
                ctx.set_option("scan_config", None)
                line += f"  B={nb}: {res[None][1]} {res[None][0]:.3f} | tp {res['tp'][0]:.3f} ms"
            print(line, flush=True)
