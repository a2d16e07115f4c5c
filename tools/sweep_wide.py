#!/usr/bin/env python3
"""GPU: the latency layout in its two forms (celerite_wide_kernel with register copies of the inputs vs the lean
celerite_wide2_kernel with slot threads) and, where it applies, the windowed kernel: ms per resident launch at N = 8192, random
(a, b, c, d) as in the reference's benchmark grid (benchmark/benchmarks.jl:74-91), B = 1 and 256; plus one CPU core (oracle).
usage: python tools/sweep_wide.py [J ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
from oracle import oracle as O

N = int(os.environ.get("N", 8192))
JS = [int(a) for a in sys.argv[1:]] or [8, 16, 20, 24, 31, 32, 40, 47, 48, 56, 64, 71]
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
rng = np.random.default_rng(5)


def med_ms(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


print("J rows B  wide_ms wide2_ms block_ms  cpu1_ms  maxrel(wide2 vs oracle)")
for J in JS:
    abcd = rng.random((256, J, 4)); abcd[:, :, 0] *= 5
    A, Bc = np.ascontiguousarray(abcd[:, :, 0]), np.ascontiguousarray(abcd[:, :, 1])
    C, Dd = np.ascontiguousarray(abcd[0, :, 2]), np.ascontiguousarray(abcd[0, :, 3])
    ds = pj.Dataset(t, y, yerr, ctx); ds.prepare(C, Dd)
    dA = torch.from_numpy(A).to(dev); dB = torch.from_numpy(Bc).to(dev)
    dout = torch.empty(256, dtype=torch.float64, device=dev); dst = torch.zeros(256, dtype=torch.int32, device=dev)
    t0 = time.perf_counter(); ref0 = O.logl(A[0], Bc[0], C, Dd, t, y, yerr); cpu = (time.perf_counter() - t0) * 1e3
    ref = O.logl_batch(A[:8], Bc[:8], C, Dd, t, y, yerr, np.zeros(8), np.ones(8), nthreads=8)
    for B in (1, 256):
        go = lambda: ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), 0, 0, 0, 0, dout.data_ptr(), dst.data_ptr())
        res = {}
        for name, opts in (("wide", {"scan_config": "wide", "no_wide2": True}), ("wide2", {"scan_config": "wide", "wide2": True}), ("block", {"scan_config": "block"})):
            for k, v in opts.items(): ctx.set_option(k, v)
            try:
                res[name] = med_ms(go)
                if name == "wide2":
                    got = dout[:min(B, 8)].cpu().numpy()
                    err = float(np.nanmax(np.abs(got - ref[:len(got)]) / np.abs(ref[:len(got)])))
            except Exception:
                res[name] = float("nan")
            for k in opts: ctx.set_option(k, None)
        print(f"{J:3d} {2*J:4d} {B:4d} {res['wide']:8.3f} {res['wide2']:8.3f} {res['block']:8.3f} {cpu:8.2f}  {err:.1e}", flush=True)
    ds.close()
