import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
pj._lib.lib()
print([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l and 'r-xp' in l])
print([l.split()[-1] for l in open('/proc/self/maps') if 'libhsa-runtime' in l and 'r-xp' in l])
N, B, J = 2000, 4096, 20
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
mu, nu = th[:, 5].copy(), th[:, 4].copy()
dev = torch.device('cuda', 0)
for mode in ("own_stream", "torch_stream"):
    stream = torch.cuda.current_stream(dev)
    ctx = pj.Context(0) if mode == "own_stream" else pj.Context(0, stream=stream.cuda_stream)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    host, hst = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True)
    ds.prepare(C, Dd, np.zeros(J, np.int32))
    dA = torch.from_numpy(A).to(dev); dB = torch.from_numpy(Bc).to(dev)
    dmu = torch.from_numpy(mu).to(dev); dnu = torch.from_numpy(nu).to(dev)
    dout = torch.full((B,), -1.0, dtype=torch.float64, device=dev); dst = torch.full((B,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ds.logl_batch_dev(B, dA.data_ptr(), dB.data_ptr(), dmu.data_ptr(), dnu.data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
    ctx.synchronize(); torch.cuda.synchronize()
    o = dout.cpu().numpy(); s = dst.cpu().numpy()
    print(mode, "stream handle", stream.cuda_stream, "match host API:", np.array_equal(o, host), "status==host:", np.array_equal(s, hst),
          "n(-1 untouched)=", int((o == -1.0).sum()), "n status -7:", int((s == -7).sum()), "host ok frac", (hst == 0).mean())
    bad = np.where(o != host)[0]
    print("  first mismatches:", bad[:10].tolist(), o[bad[:3]], host[bad[:3]])
