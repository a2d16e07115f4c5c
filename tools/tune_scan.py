#!/usr/bin/env python3
"""GPU-box tuning helper: time the scan kernel for several kernel configurations / batch sizes in ONE
process (interleaved rounds, median), same data as bench.py.  usage: tune_scan.py [B ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj

N, J = int(os.environ.get("N", 10000)), int(os.environ.get("J", 20))
basis = os.environ.get("BASIS", "SHO")
Bs = [int(x) for x in sys.argv[1:]] or [4096]
cfgs = os.environ.get("CFGS", "rpl3_cbr2_nsrc7,rpl3_cbr2_nsrc8,rpl3_cbr1_nsrc14,rpl3_cbr4_nsrc4,rpl4_cbr4_nsrc4").split(",")
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
for B in Bs:
    th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
    real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    ds.prepare(C, Dd, real.astype(np.int32))
    dA = torch.from_numpy(A).to(dev); dB = torch.from_numpy(Bc).to(dev)
    dmu = torch.from_numpy(th[:, 5].copy()).to(dev); dnu = torch.from_numpy(th[:, 4].copy()).to(dev)
    dout = torch.empty(B, dtype=torch.float64, device=dev)
    times = {c: [] for c in cfgs}
    outs = {}
    for rnd in range(4):
        for c in cfgs:
            ctx.set_option("scan_config", c)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), 0)
            e1.record(stream); e1.synchronize()
            if rnd: times[c].append(e0.elapsed_time(e1))
            outs[c] = dout.cpu().numpy().copy()
    ref = outs[cfgs[0]]
    for c in cfgs:
        ms = float(np.median(times[c]))
        dev_ = np.nanmax(np.abs(outs[c] - ref) / np.abs(ref))
        print(f"B={B:6d} {basis} {c:18s} {ms:9.3f} ms  {B / ms * 1e3:10.0f} evals/s   maxrel vs {cfgs[0]}: {dev_:.1e}", flush=True)
