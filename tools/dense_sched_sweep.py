#!/usr/bin/env python3
"""GPU: the two knobs of dense_step_kernel's bulk schedule (options dense_pair_tiles, dense_half_tile_limit, dense_no_pairs) at N = 4096, J = 40: event-timed factorisation, median of 9."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
t, y, yerr = bench.synth_series(10_000)
N = int(os.environ.get("N", 4096))
tt, yy, ee = t[:N], y[:N], yerr[:N]
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), 1 / (tt[-1] - tt[0]), 1 / np.min(np.diff(tt)) / 2, 40, 1.0, basis_function="SHO")
mu = float(np.mean(yy))
ctx = pj.Context(0)
def timed():
    ctx.dense_nll(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)
    fac = []
    for _ in range(9):
        v, info, p3 = ctx.dense_nll_timed(R.a, R.b, R.c, R.d, tt, yy - mu, ee ** 2)
        fac.append(p3["factor_ms"])
    return float(np.median(fac)), v
base, v0 = timed()
print(f"default: {base:.4f} ms  nll {v0!r}")
for pt in (0, 100, 200, 250, 300, 350, 400, 500, 700, 900, 1400, 100000):
    ctx.set_option("dense_pair_tiles", str(pt)); ms, v = timed(); print(f"pair_tiles {pt}: {ms:.4f} ms  same value {v == v0 or abs(v - v0) < 1e-9 * abs(v0)}")
ctx.set_option("dense_pair_tiles", None)
for hl in (0, 256, 512, 1024, 2048, 100000):
    ctx.set_option("dense_half_tile_limit", str(hl)); ms, v = timed(); print(f"half_tile_limit {hl}: {ms:.4f} ms")
ctx.set_option("dense_half_tile_limit", None)
base2, _ = timed(); print(f"default again: {base2:.4f} ms")
