#!/usr/bin/env python3
"""GPU: the three-step form of the scan (context option "win3", shapes up to three rows per lane, shared table) against the two-step
form: parity vs the oracle for series lengths of every remainder mod 3, and interleaved timings at B = 4096, N = 1e4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, pioran_jl_amd as pj
from oracle import oracle as O
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
rng = np.random.default_rng(1)
worst = 0.0
for J in (2, 5, 8, 12, 15, 20, 23):
    for N in (1, 2, 3, 4, 5, 6, 7, 100, 101, 102):
        B = 37
        t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.0, 3.0, J); mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        if J == 12: Bc[:, 7:] = 0.0; Dd[7:] = 0.0          # some one-row terms: unpaired layout
        ds = pj.Dataset(t, y, s2, ctx)
        ctx.set_option("no_block", True); ctx.set_option("no_wide", True); ctx.set_option("win3", True)
        got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True)
        ctx.set_option("win3", False); ctx.set_option("no_block", False); ctx.set_option("no_wide", False)
        ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=8)
        e = float(np.max(np.abs(got - ref) / np.abs(ref))); worst = max(worst, e)
        assert e < 1e-10 and (st == 0).all(), (J, N, e)
print("three-step form vs oracle: worst relative deviation", worst)
N, B = 10_000, 4096
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
JS = [int(a) for a in sys.argv[1:]] or [20, 10, 15, 23]
for J in JS:
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function="SHO")
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd)
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(B, dtype=torch.float64, device=dev)
    go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), 0)
    res = {False: [], True: []}; vals = {}
    for rep in range(4):
        for w3 in (False, True):
            ctx.set_option("win3", w3)
            go(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream); go(); go(); e1.record(stream); e1.synchronize(); res[w3].append(e0.elapsed_time(e1) / 2)
            vals[w3] = dout.clone()
    ctx.set_option("win3", False)
    ok = torch.isfinite(vals[False]) & torch.isfinite(vals[True])
    rel = float(((vals[True][ok] - vals[False][ok]).abs() / vals[False][ok].abs()).max())
    print(f"SHO-{J}: two-step {np.median(res[False]):.3f} ms, three-step {np.median(res[True]):.3f} ms, ratio {np.median(res[True]) / np.median(res[False]):.4f}, max rel diff {rel:.1e}", flush=True)
