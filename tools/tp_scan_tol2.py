#!/usr/bin/env python3
"""GPU: the scan's acceptance threshold on the WIDE models (DRWCelerite-20: 60 rows, SHO-28: 56, SHO-20: 40), 96 prior draws each at N = 1e4, B = 1: the value with the
check switched off (tp_scan_tol huge), with the walk forced behind every scan (tiny), both against the oracle; and the smallest threshold of a ladder at which the draw is
accepted (seen in the time of the call: the walk costs milliseconds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
N = 10000
t, y, yerr = bench.synth_series(N)
s2 = yerr ** 2
nd = 96
th, f_min, f_max = bench.synth_theta(nd, t, y, seed=4321)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
ladder = (1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1)
for basis, nc in (("DRWCelerite", 20), ("SHO", 28), ("SHO", 20), ("SHO", 8)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=16, return_status=True)
    ds = pj.Dataset(t, y, s2, ctx)
    ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 1)
    call = lambda i: ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])[0]
    ctx.set_option("tp_scan_tol", 1e30); call(1); t0 = time.perf_counter(); call(1); fast = time.perf_counter() - t0
    rows = []
    for i in range(nd):
        if rst[i]: continue
        ctx.set_option("tp_scan_tol", 1e30); v_scan = call(i)
        ctx.set_option("tp_scan_tol", 1e-300); v_walk = call(i)
        need = None
        for tol in ladder:
            ctx.set_option("tp_scan_tol", tol); call(i)
            t0 = time.perf_counter(); call(i); dt = time.perf_counter() - t0
            if dt < 1.6 * fast: need = tol; break
        rows.append((i, abs(v_scan - ref[i]) / abs(ref[i]), abs(v_walk - ref[i]) / abs(ref[i]), need))
    ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_scan_tol", None)
    es = np.array([r[1] for r in rows]); ew = np.array([r[2] for r in rows])
    print(f"# {basis}-{nc}: {len(rows)} positive definite draws; scan alone: max rel err {es.max():.1e} (median {np.median(es):.1e}); walk: max {ew.max():.1e} (median {np.median(ew):.1e})")
    for tol in ladder:
        sel = [r for r in rows if r[3] is not None and r[3] <= tol]
        rest = [r for r in rows if not (r[3] is not None and r[3] <= tol)]
        print(f"   threshold {tol:g}: {len(sel)} accepted, {len(rest)} to the walk; worst scan-alone error among the accepted {max([r[1] for r in sel], default=0):.1e}, among the rejected {max([r[1] for r in rest], default=0):.1e}", flush=True)
