import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
N=10000; nch=4096
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(nch, t, y, seed=4321)
basis = "DRWCelerite" if "drw" in sys.argv[1:] else "SHO"      # DRWCelerite-20: 60 rows, four block columns
A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3], basis_function=basis)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
ctx.set_option("scan_config", "tile")
cd = "cd" in sys.argv[1:]      # (d/d(c, d) of the shared (c, d) as well: the CD instantiations, round 6)
for _ in range(3): g = ds.logl_grad(A, Bc, C, Dd, mu=th[:, 5].copy(), nu=th[:, 4].copy(), cd_grad=cd)
print(pj._lib.lib().pioran_celerite_config_name(-1).decode())
