#!/usr/bin/env python3
"""Small-batch latency of the celerite scan (BASELINE config 2 and the MCMC-walker regime): ms per call of the
HBM-resident batch entry for B = 1 .. 1024 at N = 1e4: the default choice (windowed kernel up to 512 .. 1024 draws), the latency
layout (option "no_block") and the throughput layouts ("no_block" + "no_wide"), plus the scalar drop-in `logl`."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
from oracle import oracle as O

N, J = 10_000, int(os.environ.get("J", "20"))
basis = os.environ.get("BASIS", "SHO")
t, y, yerr = bench.synth_series(N)
Bmax = 1024
th, f_min, f_max = bench.synth_theta(Bmax, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
mu, nu = th[:, 5].copy(), th[:, 4].copy()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
real_term = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
ds.prepare(C, Dd, real_term.astype(np.int32))
dA = torch.from_numpy(np.ascontiguousarray(A)).to(dev); dB = torch.from_numpy(np.ascontiguousarray(Bc)).to(dev)
dmu = torch.from_numpy(mu).to(dev); dnu = torch.from_numpy(nu).to(dev)
dout = torch.empty(Bmax, dtype=torch.float64, device=dev); dst = torch.zeros(Bmax, dtype=torch.int32, device=dev)
ref = O.logl_batch(A[:16], Bc[:16], C, Dd, t, y, yerr ** 2, mu[:16], nu[:16], nthreads=8)

def run(B, reps=5):
    ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    got = dout[:min(B, 16)].cpu().numpy()
    err = float(np.nanmax(np.abs(got - ref[:len(got)]) / np.abs(ref[:len(got)])))
    return float(np.median(ts)) * 1e3, err

res = {"workload": f"N={N}, {basis}-{J}, shared (c,d) table, HBM-resident inputs", "ms_per_call": {}}
for mode in ("default", "latency_layout", "throughput_layouts"):
    ctx.set_option("no_block", mode != "default")
    ctx.set_option("no_wide", mode == "throughput_layouts")
    row = {}
    for B in (1, 4, 16, 64, 128, 256, 512, 1024):
        ms, err = run(B)
        row[str(B)] = round(ms, 3)
        assert err < 1e-8, (mode, B, err)
    res["ms_per_call"][mode] = row
    ctx.set_option("no_wide", False); ctx.set_option("no_block", False)
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), f_min, f_max, J, 1.0, basis_function=basis)
for n in (10_000, 1000):
    pj.logl(R.a, R.b, R.c, R.d, t[:n], y[:n], yerr[:n] ** 2, ctx=ctx)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pj.logl(R.a, R.b, R.c, R.d, t[:n], y[:n], yerr[:n] ** 2, ctx=ctx); ts.append(time.perf_counter() - t0)
    res[f"scalar_logl_ms_N{n}"] = round(float(np.median(ts)) * 1e3, 3)
# one CPU core of this host on the same single evaluation (oracle = the reference's algorithm and layout)
for n in (10_000, 1000):   # BASELINE configs[0] is the N = 1000 single evaluation on the CPU path
    O.logl(R.a, R.b, R.c, R.d, t[:n], y[:n], yerr[:n] ** 2)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); O.logl(R.a, R.b, R.c, R.d, t[:n], y[:n], yerr[:n] ** 2); ts.append(time.perf_counter() - t0)
    res[f"cpu_one_core_ms_N{n}"] = round(float(np.median(ts)) * 1e3, 3)
print(json.dumps(res))
