#!/usr/bin/env python3
"""GPU: the two-step scan with the series in segments (context option "segments") against the same kernel in one piece — the
arithmetic is the same, so the results must agree to the bit — and the launch time of both, by batch size and basis.
usage: python tools/seg_check.py [segments ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, pioran_jl_amd as pj
segs = [int(v) for v in sys.argv[1:]] or [2, 4]
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
bad = 0
for N, J, basis in ((10_000, 20, "SHO"), (9_999, 20, "SHO"), (10_000, 20, "DRWCelerite"), (10_000, 30, "SHO"), (3_001, 12, "SHO"), (777, 40, "DRWCelerite")):
    t, y, yerr = bench.synth_series(N)
    BS = (1536, 2048, 4096, 8192) if N == 10_000 and J == 20 else (4096,)
    th, f_min, f_max = bench.synth_theta(max(BS), t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
    Jc = A.shape[1]
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, ((Dd == 0.0) & (Bc == 0.0).all(axis=0)).astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    for B in BS:
        res = {}
        for sg in [0] + segs:
            ctx.set_option("segments", str(sg))
            dout = torch.full((B,), np.nan, dtype=torch.float64, device=dev)
            go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), 0)
            go(); torch.cuda.synchronize(); ts = []
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream); go(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
            res[sg] = (dout.cpu().numpy().copy(), float(np.median(ts)), pj._lib.lib().pioran_celerite_config_name(0).decode())
        base = res[0][0]
        line = f"N={N} J={Jc} {basis:11s} B={B:5d} [{res[0][2]}]: one piece {res[0][1]:7.3f} ms"
        for sg in segs:
            same = np.array_equal(res[sg][0], base, equal_nan=False)
            bad += 0 if same else 1
            line += f" | {sg} segments {res[sg][1]:7.3f} ms {'same bits' if same else 'DIFFERENT (%d draws, nan %d)' % ((res[sg][0] != base).sum(), np.isnan(res[sg][0]).sum())}"
        print(line, flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
