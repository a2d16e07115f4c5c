#!/usr/bin/env python3
"""GPU: the draws bench.py's `single_evaluation_long_series` uses — theta drawn for the N = 1e4 series (seed 4321), evaluated on the N = 65536 series with ITS frequency range — through
the scan alone (check off) against the oracle, with the smallest threshold of a ladder at which the check accepts each (from the time of the call).  One of them (draw 0, DRWCelerite-20)
came back 5e-6 off AND accepted under the first form of the check (discrepancy relative to the state's largest entry, threshold 1e-5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 48
t0_, y0_, e0_ = bench.synth_series(10000)
theta, _, _ = bench.synth_theta(4096, t0_, y0_, seed=4321)
ladder = (1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, 10.0)
for NL in (65536, 30000):
    tL, yL, eL = bench.synth_series(NL)
    fm, fM = 1.0 / (tL[-1] - tL[0]), 1.0 / (2 * np.min(np.diff(tL)))
    s2 = eL ** 2
    for basis in ("DRWCelerite", "SHO"):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, theta[:nd, :3], fm, fM, 20, theta[:nd, 3], basis_function=basis)
        mu, nu = theta[:nd, 5].copy(), theta[:nd, 4].copy()
        ref, rst = O.logl_batch(A, Bc, C, Dd, tL, yL, s2, mu, nu, nthreads=16, return_status=True)
        ds = pj.Dataset(tL, yL, s2, ctx)
        call = lambda i: ds.logl_batch(A[i:i + 1], Bc[i:i + 1], C, Dd, mu=mu[i:i + 1], nu=nu[i:i + 1], return_status=True)
        ctx.set_option("tp_scan_tol", 1e30); call(1); ts = []
        for _ in range(5):
            tt = time.perf_counter(); call(1); ts.append(time.perf_counter() - tt)
        fast = min(ts)
        rows = []
        for i in range(nd):
            if rst[i]: continue
            ctx.set_option("tp_scan_tol", 1e30); vs, _ = call(i)
            es = abs(vs[0] - ref[i]) / abs(ref[i])
            need = float("inf")
            for tol in ladder:
                ctx.set_option("tp_scan_tol", tol); call(i); tt = time.perf_counter(); call(i)
                if time.perf_counter() - tt < 1.8 * fast: need = tol; break
            ctx.set_option("tp_scan_tol", None)
            v, _ = call(i)
            rows.append((i, es, need, abs(v[0] - ref[i]) / abs(ref[i])))
        bad = [r for r in rows if r[1] > 1e-9]
        print(f"{basis}-20 N={NL}: {len(rows)} positive definite draws; scan alone: max {max(r[1] for r in rows):.1e}, {sum(r[1] > 1e-8 for r in rows)} above 1e-8; product path max {max(r[3] for r in rows):.1e}")
        for r in bad: print(f"     draw {r[0]}: scan alone {r[1]:.1e}, accepted from threshold {r[2]:g} on, product path {r[3]:.1e}")
        for tol in ladder:
            acc = [r for r in rows if r[2] <= tol]
            print(f"     threshold {tol:g}: {len(acc)} accepted, worst scan-alone error among them {max([r[1] for r in acc], default=0):.1e}", flush=True)
