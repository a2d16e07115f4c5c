#!/usr/bin/env python3
"""GPU: the time-parallel family's boundary phase as a scan over the segments' elements (tp_combine_kernel, option tp_scan = 1, round 6) against the sequential
walk (tp_scan = 0) and the oracle: accuracy on small shapes (scan forced at segment counts that are and are not powers of two), then N = 1e4 timings per
segment count.  usage: python tools/ab_tp_scan.py [time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(11)
worst = 0.0
for J, N, B, nreal, nseg in ((4, 300, 1, 0, 2), (4, 300, 2, 0, 3), (8, 500, 1, 0, 5), (8, 640, 2, 4, 8), (12, 700, 1, 0, 17), (16, 900, 1, 0, 16), (20, 1000, 2, 0, 7), (20, 2000, 1, 0, 33),
                             (24, 1500, 1, 0, 12), (21, 800, 1, 18, 9), (20, 4000, 1, 0, 128), (10, 600, 3, 4, 6)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.1, 3.0, J)
    if nreal: Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
    ds = pj.Dataset(t, y, s2, ctx)
    ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=4)
    out = {}
    for mode in (1, 0):
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_segments", nseg); ctx.set_option("tp_scan", mode)
        got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True); k = name()
        out[mode] = got
    ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0); ctx.set_option("tp_scan", -1)
    e1 = float(np.max(np.abs(out[1] - ref) / np.abs(ref))); e0 = float(np.max(np.abs(out[0] - ref) / np.abs(ref)))
    worst = max(worst, e1)
    print(f"J={J} N={N} B={B} nreal={nreal} nseg={nseg}: [{k}] scan vs oracle {e1:.2e}, walk vs oracle {e0:.2e}, scan vs walk {float(np.max(np.abs(out[1] - out[0]) / np.abs(out[0]))):.2e} status {st.tolist()}", flush=True)
print("worst", worst)
if len(sys.argv) > 1:
    N = 10000
    t, y, yerr = bench.synth_series(N)
    th, f_min, f_max = bench.synth_theta(8, t, y, seed=4321)
    for basis, nc in (("SHO", 20), ("SHO", 24), ("SHO", 16), ("SHO", 12), ("SHO", 8), ("DRWCelerite", 10), ("SHO", 4)):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
        ds = pj.Dataset(t, y, yerr ** 2, ctx)
        for nb in (1, 2):
            ref = O.logl_batch(A[:nb], Bc[:nb], C, Dd, t, y, yerr ** 2, th[:nb, 5].copy(), th[:nb, 4].copy(), nthreads=4)
            for mode, segs in ((0, [0]), (1, [0, 16, 32, 64, 128]), (-1, [0])):
                for sg in segs:
                    ctx.set_option("scan_config", "tp" if mode >= 0 else None); ctx.set_option("tp_segments", sg); ctx.set_option("tp_scan", mode)
                    try:
                        got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy())
                    except Exception as ex:
                        print(f"{basis}-{nc} B={nb} scan={mode} segs={sg}: {ex}"); continue
                    ts = []
                    for _ in range(7):
                        t0 = time.perf_counter(); got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy()); ts.append(time.perf_counter() - t0)
                    e = float(np.max(np.abs(got - ref) / np.abs(ref)))
                    print(f"{basis}-{nc} N={N} B={nb} [{name()}] scan={mode} segs={sg}: {min(ts) * 1e3:.3f} ms per call (host entry, PCIe included); max rel vs oracle {e:.2e}", flush=True)
            ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0); ctx.set_option("tp_scan", -1)
