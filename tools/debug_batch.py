"""Debug helper (GPU box): compare batch results vs the oracle for different batch sizes / configs."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, B, J = int(os.environ.get("N", 10000)), 4096, 20
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
mu, nu = th[:, 5].copy(), th[:, 4].copy()
S = 256
ref, rst = O.logl_batch(A[:S], Bc[:S], C, Dd, t, y, yerr ** 2, mu[:S], nu[:S], nthreads=64, return_status=True)
ctx = pj.Context(0)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
def run(Bn, tag):
    got, st = ds.logl_batch(A[:Bn], Bc[:Bn], C, Dd, mu=mu[:Bn], nu=nu[:Bn], return_status=True)
    m = min(Bn, S)
    rel = np.abs(got[:m] - ref[:m]) / np.abs(ref[:m])
    bad = np.where(~(rel < 1e-8))[0]
    print(f"{tag}: B={Bn} status_ok={np.mean(st==0):.3f} bad(of first {m})={len(bad)} first bad idx={bad[:24].tolist()}")
    return got, st
for Bn in (8, 16, 64, 256, 512, 1024, 2048, 4096):
    got, st = run(Bn, "default")
g1, s1 = run(4096, "again")
print("repeatable:", np.array_equal(got, g1, equal_nan=True))
bad = np.where(st != 0)[0]
print("status!=0 idx (first 40):", bad[:40].tolist())
print("idx mod 8 histogram of bad:", np.bincount(bad % 8, minlength=8).tolist())
print("block (idx//8) mod 8 hist:", np.bincount((bad // 8) % 8, minlength=8).tolist())
for name in ("rpl3_cbr1_nsrc14", "rpl3_cbr4_nsrc4", "rpl3_cbr2_nsrc8"):
    os.environ["PIORAN_SCAN_CONFIG"] = name
    run(4096, name)
os.environ.pop("PIORAN_SCAN_CONFIG")
os.environ["PIORAN_FORCE_FALLBACK"] = "1"
run(1024, "fallback")
