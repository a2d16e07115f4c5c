#!/usr/bin/env python3
"""GPU: the scan's combinations with their operands from global memory (tp_combine_lean_kernel: the only form that fits 49 .. 64 state rows) against
tp_combine_kernel at up to 48 rows (option tp_scan_lean), against the walk and the oracle at 56 / 64 rows; then DRWCelerite-20 / SHO-28 / SHO-32 timings at
N = 1e4 .. 65536 per segment count.  usage: python tools/ab_tp_scan_lean.py [time]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rng = np.random.default_rng(12)
rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
for J, N, B, nreal, nseg in ((4, 300, 1, 0, 3), (8, 640, 2, 4, 8), (12, 700, 1, 0, 17), (20, 1000, 2, 0, 7), (24, 1500, 1, 0, 12), (21, 800, 1, 18, 9), (20, 4000, 1, 0, 128),
                             (28, 900, 1, 0, 5), (32, 1200, 2, 0, 16), (30, 1500, 1, 0, 33), (40, 2000, 2, 20, 64), (36, 700, 1, 9, 2), (31, 4100, 1, 3, 256)):
    t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.1, 3.0, J)
    if nreal: Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
    ds = pj.Dataset(t, y, s2, ctx)
    ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=4)
    out = {}
    for label, mode, lean in (("lean", 1, 1), ("lds", 1, 0), ("walk", 0, 0)):
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_segments", nseg); ctx.set_option("tp_scan", mode); ctx.set_option("tp_scan_lean", lean)
        got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True); k = name()
        out[label] = got
    ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0); ctx.set_option("tp_scan", -1); ctx.set_option("tp_scan_lean", 0)
    print(f"rows={2 * J - nreal} N={N} B={B} nseg={nseg}: [{k}] lean vs oracle {rel(out['lean'], ref):.2e}, lds-form vs oracle {rel(out['lds'], ref):.2e}, walk vs oracle {rel(out['walk'], ref):.2e}, "
          f"lean vs lds-form {rel(out['lean'], out['lds']):.2e} status {st.tolist()}", flush=True)
if len(sys.argv) > 1:
    for N in (10000, 16384, 65536):
        t, y, yerr = bench.synth_series(N)
        th, f_min, f_max = bench.synth_theta(8, t, y, seed=4321)
        for basis, nc in (("DRWCelerite", 20), ("SHO", 28), ("SHO", 32), ("SHO", 20)):
            A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
            ds = pj.Dataset(t, y, yerr ** 2, ctx)
            nb = 1
            ref = O.logl_batch(A[:nb], Bc[:nb], C, Dd, t, y, yerr ** 2, th[:nb, 5].copy(), th[:nb, 4].copy(), nthreads=4)
            for label, cfg, mode, lean, segs in (("serial", None, -1, 0, [0]), ("walk", "tp", 0, 0, [0]), ("scan", "tp", 1, 0, [0, 32, 64, 128, 256]), ("scan-lean", "tp", 1, 1, [0] if nc == 20 and basis == "SHO" else [])):
                for sg in segs:
                    ctx.set_option("no_tp", label == "serial"); ctx.set_option("scan_config", cfg); ctx.set_option("tp_segments", sg); ctx.set_option("tp_scan", mode); ctx.set_option("tp_scan_lean", lean)
                    try:
                        got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy())
                    except Exception as ex:
                        print(f"{basis}-{nc} {label} segs={sg}: {ex}"); continue
                    ts = []
                    for _ in range(7):
                        t0 = time.perf_counter(); got = ds.logl_batch(A[:nb], Bc[:nb], C, Dd, mu=th[:nb, 5].copy(), nu=th[:nb, 4].copy()); ts.append(time.perf_counter() - t0)
                    print(f"{basis}-{nc} N={N} B={nb} [{name()}] {label} segs={sg}: {min(ts) * 1e3:.3f} ms per call (host entry, PCIe included); max rel vs oracle {rel(got, ref):.2e}", flush=True)
            ctx.set_option("no_tp", False); ctx.set_option("scan_config", None); ctx.set_option("tp_segments", 0); ctx.set_option("tp_scan", -1); ctx.set_option("tp_scan_lean", 0)
