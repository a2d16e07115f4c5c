#!/usr/bin/env python3
"""GPU: the product path for one draw of a long series (automatic choice: the time-parallel scan with its check and the serial-chain repair pass) on 512 prior draws per model at
N = 1e4: largest deviation from the oracle over the positive definite draws, how many were repaired (seen in the time of the call), and what the scan alone (check off) would have given."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
N = 10000
t, y, yerr = bench.synth_series(N)
s2 = yerr ** 2
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 512
th, f_min, f_max = bench.synth_theta(nd, t, y, seed=2024)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
for basis, nc in (("DRWCelerite", 20), ("DRWCelerite", 10), ("SHO", 20), ("SHO", 32), ("SHO", 8)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=16, return_status=True)
    ds = pj.Dataset(t, y, s2, ctx)
    call = lambda i: ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1], return_status=True)
    call(1); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); call(1); ts.append(time.perf_counter() - t0)
    fast = min(ts)
    err, err_scan, rep, bad_status = [], [], 0, 0
    for i in range(nd):
        call(i); t0 = time.perf_counter(); v, st = call(i); dt = time.perf_counter() - t0
        k = name()
        if rst[i] or st[0]:
            bad_status += int(bool(rst[i]) != bool(st[0])); continue
        rep += dt > 1.8 * fast
        err.append(abs(v[0] - ref[i]) / abs(ref[i]))
        ctx.set_option("tp_scan_tol", 1e30); vs, _ = call(i); ctx.set_option("tp_scan_tol", None)
        err_scan.append(abs(vs[0] - ref[i]) / abs(ref[i]))
    err, err_scan = np.array(err), np.array(err_scan)
    print(f"{basis}-{nc} [{k}] {fast * 1e3:.3f} ms per call: {len(err)} positive definite draws of {nd} (status disagrees with the oracle's on {bad_status}); product path: max rel err {err.max():.1e}, "
          f"{(err > 1e-8).sum()} above 1e-8, {(err > 1e-9).sum()} above 1e-9; {rep} repaired on the serial chain; the scan alone (check off): max {err_scan.max():.1e}, {(err_scan > 1e-8).sum()} above 1e-8", flush=True)
