#!/usr/bin/env python3
"""celerite_tile.hip (windowed form, one draw per wavefront) against the automatic kernels and the oracle: accuracy and time per launch.
usage: python tools/ab_tile.py [N] [B] [cases: sho20,drw20,sho40,sho12,...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Bs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096]
BM = max(Bs)
cases = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sho20", "drw20", "sho40"]
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
th, f_min, f_max = bench.synth_theta(BM, t, y, seed=4321)
def med_ms(f, reps=3):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
name = lambda i: pj._lib.lib().pioran_celerite_config_name(i).decode()
for case in cases:
    basis = "DRWCelerite" if case.startswith("drw") else "SHO"
    nc = int(case[3:])
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    J = A.shape[1]
    real_term = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    R = int(2 * J - real_term.sum())
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real_term.astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(BM, dtype=torch.float64, device=dev); dst = torch.zeros(BM, dtype=torch.int32, device=dev)
    no = min(min(Bs), 64)
    ref, rst = O.logl_batch(A[:no], Bc[:no], C, Dd, t, y, yerr ** 2, th[:no, 5].copy(), th[:no, 4].copy(), nthreads=8, return_status=True)
    ok = rst == 0
    for B in Bs:
        go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
        ctx.set_option("scan_config", "")
        ms0 = med_ms(go); g0 = dout[:B].cpu().numpy().copy(); s0 = dst[:B].cpu().numpy().copy(); k0 = name(-1) + ":" + name(0)
        ctx.set_option("scan_config", "tile")
        ms1 = med_ms(go); g1 = dout[:B].cpu().numpy().copy(); s1 = dst[:B].cpu().numpy().copy(); k1 = name(-1)
        ctx.set_option("scan_config", "")
        F = (N - 1) * (5.5 * R * R + 18 * R) * B
        e0 = np.max(np.abs(g0[:no][ok] - ref[ok]) / np.abs(ref[ok])); e1 = np.max(np.abs(g1[:no][ok] - ref[ok]) / np.abs(ref[ok]))
        both = (s0 == 0) & (s1 == 0)
        dd = np.max(np.abs(g0[both] - g1[both]) / np.abs(g0[both]))
        print(f"{case} N={N} B={B} rows={R}: auto [{k0}] {ms0:.2f} ms = {B / ms0:.1f} k/s ({F / ms0 / 1e9 / 78.6:.3f}); tile [{k1}] {ms1:.2f} ms = {B / ms1:.1f} k/s ({F / ms1 / 1e9 / 78.6:.3f}); "
              f"max rel vs oracle ({ok.sum()} draws) auto {e0:.1e} tile {e1:.1e}; tile vs auto over {both.sum()} draws {dd:.1e}; status differs on {(s0 != s1).sum()}", flush=True)
