#!/usr/bin/env python3
"""GPU: resident launch time of the headline workload (N = 1e4, SHO-20) by batch size — whole passes of 2048 wavefronts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, pioran_jl_amd as pj
N = 10_000
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
BS = (1536, 2048, 3072, 4096, 4200, 5000, 6144, 8192)
th, f_min, f_max = bench.synth_theta(max(BS), t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3])
ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, np.zeros(20, np.int32))
d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
dout = torch.empty(max(BS), dtype=torch.float64, device=dev)
for B in BS:
    go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), 0)
    go(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); go(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print(f"B = {B:5d}: {ms:7.3f} ms per launch, {B / ms:7.1f} evals/ms", flush=True)
