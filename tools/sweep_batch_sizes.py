#!/usr/bin/env python3
"""GPU: launch time by batch size, with and without the remainder split (capi.hip split_dispatch; option no_split), N = 1e4, resident inputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
N = 10000
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
BMAX = 8192
th, f_min, f_max = bench.synth_theta(BMAX, t, y, seed=4321)
def med_ms(f, reps=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for basis in ("SHO", "DRWCelerite"):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3], basis_function=basis)
    real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real.astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(BMAX, dtype=torch.float64, device=dev); dst = torch.zeros(BMAX, dtype=torch.int32, device=dev)
    for nb in (1024, 1536, 2048, 2500, 3072, 4096, 4200, 4608, 5000, 6144, 6500, 8192):
        go = lambda: ds.logl_batch_dev(nb, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
        ctx.set_option("no_split", True); m0 = med_ms(go); k0 = pj._lib.lib().pioran_celerite_config_name(-1).decode(); r0 = dout[:nb].cpu().numpy().copy()
        ctx.set_option("no_split", False); m1 = med_ms(go); k1 = pj._lib.lib().pioran_celerite_config_name(-1).decode(); r1 = dout[:nb].cpu().numpy().copy()
        ok = np.isfinite(r0) & np.isfinite(r1)
        print(f"{basis}-20 B = {nb:5d}: one launch {m0:7.3f} ms ({nb / m0:6.1f} evals/ms, {k0}) | split {m1:7.3f} ms ({nb / m1:6.1f} evals/ms, {k1}) | "
              f"max rel diff {np.max(np.abs(r0[ok] - r1[ok]) / np.abs(r0[ok])):.1e}", flush=True)
    ds.close()
