#!/usr/bin/env python3
"""GPU: the windowed kernel with its pair table E in one LDS buffer (0), two (1) or read from global memory (2: 61 instead of 102 KB of LDS
per workgroup at J = 20, so two workgroups share a CU) — context option "block_emode"; resident launches, N = 1e4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
N = 10_000
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
BS = (64, 256, 400, 512, 768, 1024)
th, f_min, f_max = bench.synth_theta(max(BS), t, y, seed=4321)
def med_ms(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for basis, J in (("SHO", 20), ("SHO", 10), ("SHO", 5), ("DRWCelerite", 10)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
    real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real.astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(max(BS), dtype=torch.float64, device=dev)
    ctx.set_option("scan_config", "block")
    print(f"# {basis}-{J} ({int(2 * len(C) - real.sum())} rows): B | E in one LDS buffer | two | global memory   (ms per launch); max rel diff vs one buffer")
    for B in BS:
        go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), 0)
        ms = []; ref = None; worst = 0.0
        for em in (0, 1, 2):
            ctx.set_option("block_emode", em)
            ms.append(med_ms(go))
            got = dout[:B].clone()
            if ref is None: ref = got
            else:
                ok = torch.isfinite(ref) & torch.isfinite(got)
                worst = max(worst, float(((got[ok] - ref[ok]).abs() / ref[ok].abs()).max()))
        ctx.set_option("block_emode", None)
        print(f"{B:5d} | {ms[0]:7.3f} | {ms[1]:7.3f} | {ms[2]:7.3f} | {worst:.1e}", flush=True)
    ctx.set_option("scan_config", None)
    ds.close()
