"""GPU: a few value + gradient calls at N = 1e4 (J, B in the environment) — the workload of tools/kstats.sh runs."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
t, y, yerr = bench.synth_series(10000)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
J = int(os.environ.get("J", 40)); B = int(os.environ.get("B", 1))
th = O.synthetic_theta(B, t, y, seed=J)
A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, J, "SHO")
for _ in range(3): ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
