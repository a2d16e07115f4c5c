"""Resident launch of B draws (default 256) of SHO-20 at N = 1e4 through the automatic choice, best of 40 (ms).  For a same-box A/B through PIORAN_HIP_LIB."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
basis = sys.argv[2] if len(sys.argv) > 2 else "SHO"
N = 10000
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(max(B, 64), t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:B, :3], f_min, f_max, 20, th[:B, 3], basis_function=basis)
dev = torch.device("cuda", 0)
ctx = pj.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
real_term = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
ds.prepare(C, Dd, real_term.astype(np.int32))
dA = torch.from_numpy(np.ascontiguousarray(A)).to(dev); dB = torch.from_numpy(np.ascontiguousarray(Bc)).to(dev)
dmu = torch.from_numpy(th[:B, 5].copy()).to(dev); dnu = torch.from_numpy(th[:B, 4].copy()).to(dev)
dout = torch.empty(B, dtype=torch.float64, device=dev); dst = torch.zeros(B, dtype=torch.int32, device=dev)
best = 1e9
for _ in range(41):
    t0 = time.perf_counter()
    ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"{best * 1e3:.4f} {pj._lib.lib().pioran_celerite_config_name(-1).decode()}")
