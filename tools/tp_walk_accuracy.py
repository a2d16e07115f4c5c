#!/usr/bin/env python3
"""GPU: the time-parallel family's sequential form (boundary walk, unchecked: option tp_unchecked) against the oracle on 160 prior draws of DRWCelerite-10 / -20 at N = 1e4, beside
the serial-chain windowed kernel, the scan alone and the product path (check + repair).  Finding (late round 6): the walk alone is off by up to 8e-7 (2e-6 at 32 segments) on two
of the DRWCelerite-10 draws — a long segment's element is no better conditioned than a composite of the scan — where the windowed kernel holds 4e-10: the walk is checked and
repaired like the scan since."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
t, y, yerr = O.synthetic_series(10_000, seed=1234)
th = O.synthetic_theta(160, t, y, seed=909)
ds = pj.Dataset(t, y, yerr ** 2, ctx)
for ncomp in (10, 20):
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, ncomp, "DRWCelerite")
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, yerr ** 2, mu, nu, nthreads=16, return_status=True)
    res = {}
    for label, opts in (("walk, unchecked", {"scan_config": "tp", "tp_scan": 0, "tp_unchecked": True}), ("walk 32 segs, unchecked", {"scan_config": "tp", "tp_scan": 0, "tp_segments": 32, "tp_unchecked": True}),
                         ("walk, checked", {"scan_config": "tp", "tp_scan": 0}), ("block", {"no_tp": True}), ("scan alone", {"tp_scan_tol": 1e30}), ("product", {})):
        for k, v in opts.items(): ctx.set_option(k, v)
        out = np.array([ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1])[0] for i in range(len(th))])
        for k in opts: ctx.set_option(k, None if k in ("scan_config", "tp_scan_tol") else (-1 if k == "tp_scan" else (0 if k == "tp_segments" else False)))
        ok = rst == 0
        e = np.abs(out - ref) / np.abs(ref)
        worst = np.argsort(-np.where(ok, e, 0))[:3]
        print(f"DRWCelerite-{ncomp} {label:24s}: max rel err vs oracle {e[ok].max():.1e}; worst draws {[(int(i), float(f'{e[i]:.1e}')) for i in worst]}", flush=True)
