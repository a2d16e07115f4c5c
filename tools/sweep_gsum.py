#!/usr/bin/env python3
"""GPU: the row sums of the two-step scan with fewer ds_bpermute exchange rounds (context option "gsum": 0 = NSRC lanes +
log2(CBR) rounds, 1 = 2 NSRC lanes + one round [four DPP rows per draw only], 2 = all 16 lanes, no round).  ms per resident
launch at N = 1e4.  usage: python tools/sweep_gsum.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj

N = int(os.environ.get("N", 10_000))
BS = [int(b) for b in os.environ.get("BS", "1024,2048,4096").split(",")]
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
th, f_min, f_max = bench.synth_theta(max(BS), t, y, seed=4321)
name = lambda: pj._lib.lib().pioran_celerite_config_name(0).decode()


def med_ms(f, reps=4):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


CASES = [("DRWCelerite", 20, None), ("SHO", 20, None), ("SHO", 20, "rpl3_cbr4_nsrc4"), ("SHO", 30, None), ("SHO", 23, "rpl4_cbr4_nsrc4")]
print("basis-J rows config B | gsum=0 ms | gsum=1 ms | gsum=2 ms | max rel diff vs gsum=0")
for basis, J, cfg in CASES:
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
    real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    R = int(2 * len(C) - real.sum())
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real.astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(max(BS), dtype=torch.float64, device=dev); dst = torch.zeros(max(BS), dtype=torch.int32, device=dev)
    ctx.set_option("no_block", True); ctx.set_option("no_wide", True); ctx.set_option("scan_config", cfg)
    for B in BS:
        go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
        ms = []; ref = None; worst = 0.0
        for gs in (0, 1, 2):
            ctx.set_option("gsum", gs)
            ms.append(med_ms(go))
            got = dout[:B].clone()
            if ref is None: ref = got
            else:
                ok = torch.isfinite(ref) & torch.isfinite(got)
                worst = max(worst, float(((got[ok] - ref[ok]).abs() / ref[ok].abs()).max()))
        ctx.set_option("gsum", None)
        print(f"{basis}-{J} {R} {name()} {B:5d} | {ms[0]:7.3f} | {ms[1]:7.3f} | {ms[2]:7.3f} | {worst:.1e}", flush=True)
    for k in ("no_block", "no_wide", "scan_config"): ctx.set_option(k, None)
    ds.close()
