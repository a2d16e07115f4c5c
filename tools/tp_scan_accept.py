#!/usr/bin/env python3
"""GPU: the product path for one draw of a long series (automatic choice: the time-parallel scan with its check and the serial-chain repair pass) on prior draws per model:
largest deviation from the oracle over the positive definite draws, how many were repaired, what the scan alone (check off) would have given, and — per threshold of a ladder —
how many draws it accepts and the worst scan-alone error among them (the smallest accepting threshold of a draw is found from the time of the call: a repaired call costs
milliseconds more).  usage: tp_scan_accept.py [draws = 512] [N = 10000] [seed = 2024] [two models only]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
t, y, yerr = bench.synth_series(N)
s2 = yerr ** 2
ladder = (1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2024
th, f_min, f_max = bench.synth_theta(nd, t, y, seed=seed)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
models = (("DRWCelerite", 20), ("DRWCelerite", 10), ("SHO", 20), ("SHO", 32), ("SHO", 8)) if len(sys.argv) <= 4 else (("DRWCelerite", 20), ("SHO", 20))
for basis, nc in models:
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=16, return_status=True)
    ds = pj.Dataset(t, y, s2, ctx)
    call = lambda i: ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1], return_status=True)
    ctx.set_option("tp_scan_tol", 1e30); call(1); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); call(1); ts.append(time.perf_counter() - t0)
    fast = min(ts); ctx.set_option("tp_scan_tol", None)
    err, err_scan, need, rep, bad_status = [], [], [], 0, 0
    for i in range(nd):
        call(i); t0 = time.perf_counter(); v, st = call(i); dt = time.perf_counter() - t0
        k = name()
        if rst[i] or st[0]:
            bad_status += int(bool(rst[i]) != bool(st[0])); continue
        rep += dt > 1.8 * fast
        err.append(abs(v[0] - ref[i]) / abs(ref[i]))
        ctx.set_option("tp_scan_tol", 1e30); vs, _ = call(i)
        err_scan.append(abs(vs[0] - ref[i]) / abs(ref[i]))
        nd_ = 1.0
        for tol in ladder:                        # the smallest threshold of the ladder that accepts the draw (a repaired call costs milliseconds more)
            ctx.set_option("tp_scan_tol", tol); call(i); t0 = time.perf_counter(); call(i)
            if time.perf_counter() - t0 < 1.8 * fast: nd_ = tol; break
        need.append(nd_)
        ctx.set_option("tp_scan_tol", None)
    err, err_scan, need = np.array(err), np.array(err_scan), np.array(need)
    print(f"{basis}-{nc} N={N} seed {seed} [{k}] {fast * 1e3:.3f} ms per call: {len(err)} positive definite draws of {nd} (status disagrees with the oracle's on {bad_status}); product path: max rel err {err.max():.1e}, "
          f"{(err > 1e-8).sum()} above 1e-8, {(err > 1e-9).sum()} above 1e-9; {rep} repaired on the serial chain; the scan alone (check off): max {err_scan.max():.1e}, {(err_scan > 1e-8).sum()} above 1e-8", flush=True)
    for tol in ladder:
        acc = need <= tol
        print(f"     threshold {tol:g}: {int(acc.sum())} accepted, worst scan-alone error among them {err_scan[acc].max() if acc.any() else 0:.1e}; {int((~acc).sum())} repaired, smallest scan-alone error among those {err_scan[~acc].min() if (~acc).any() else 0:.1e}", flush=True)
