#!/usr/bin/env python3
"""Throughput of the shifted log-flux batch next to the plain batch, same size as bench.py:
N=1e4, SHO-20, B=4096, device-resident inputs.  One JSON line."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
N, B, J = 10_000, 4096, 20
t, ylog, yerr = bench.synth_series(N)
flux = np.exp(ylog); ferr = yerr * flux                       # raw flux series; models fit log(flux - c)
th, f_min, f_max = bench.synth_theta(B, t, ylog, seed=4321)
shift = np.exp(np.random.default_rng(5).uniform(np.log(1e-6), np.log(0.5 * flux.min()), B))
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
ds = pj.Dataset(t, flux, ferr ** 2, ctx); ds.prepare(C, Dd, np.zeros(J, np.int32))
dA = torch.from_numpy(A).to(dev); dB = torch.from_numpy(Bc).to(dev)
dmu = torch.from_numpy(th[:, 5].copy()).to(dev); dnu = torch.from_numpy(th[:, 4].copy()).to(dev)
dsh = torch.from_numpy(shift).to(dev); dout = torch.empty(B, dtype=torch.float64, device=dev)
def run(fn):
    ts = []
    for i in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); e1.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
ms_shift = run(lambda: ds.logl_batch_shift_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), dsh.data_ptr(), dout.data_ptr(), 0))
got = dout.cpu().numpy().copy()
ms_plain = run(lambda: ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), 0))
# parity of the shifted path on a sample, oracle as checker
from oracle import oracle as O
S = 64
ref = np.array([O.logl(A[i], Bc[i], C, Dd, t, np.log(flux - shift[i]) - th[i, 5], th[i, 4] * ferr ** 2 / (flux - shift[i]) ** 2) for i in range(S)])
ok = np.isfinite(ref)
print(json.dumps({"workload": f"N={N}, SHO-{J}, B={B}, per-draw shift transform on device", "ms_shift": ms_shift,
                  "evals_per_s_shift": B / ms_shift * 1e3, "ms_plain": ms_plain, "evals_per_s_plain": B / ms_plain * 1e3,
                  "hbm_bytes_transformed_series": 16 * B * N,
                  "max_rel_err_vs_oracle_sample": float(np.max(np.abs(got[:S][ok] - ref[ok]) / np.abs(ref[ok])))}))
