#!/usr/bin/env python3
"""GPU: the scan's check with one filter sweep (the scan's states as they are) and with two (the second corrects them: the product's form) on many draws, scalar
entry (series already mean-subtracted and scaled on the host, as bench.py's and the Julia shim's scalar calls do): per draw the error of the scan ALONE against the ORACLE and the smallest threshold of a ladder at which each measure accepts it.  For each measure and threshold: how many draws it accepts and the worst
scan-alone error among them.  usage: tp_scan_metrics.py [draws = 192] [seed = 4321] [other: DRWCelerite-15, SHO-12, SHO-24, SHO-4 instead of DRWCelerite-20, SHO-20, DRWCelerite-10, SHO-8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench, pioran_jl_amd as pj
from oracle import oracle as O
ctx = pj.Context(0)
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 192
t0_, y0_, e0_ = bench.synth_series(10000)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4321
theta, _, _ = bench.synth_theta(4096, t0_, y0_, seed=seed)
ladder = (1e-11, 1e-10, 1e-9, 3e-9, 1e-8, 3e-8, 1e-7, 1e-6, 1e-5)
ladder_raw = (1e-8, 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3, 1e-2)
allrows = {0: [], 1: []}
for NL in (10000, 30000, 65536):
    tL, yL, eL = bench.synth_series(NL)
    fm, fM = 1.0 / (tL[-1] - tL[0]), 1.0 / (2 * np.min(np.diff(tL)))
    for basis, nc in ((("DRWCelerite", 20), ("SHO", 20), ("DRWCelerite", 10), ("SHO", 8)) if len(sys.argv) <= 3 else (("DRWCelerite", 15), ("SHO", 12), ("SHO", 24), ("SHO", 4))):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, theta[:nd, :3], fm, fM, nc, theta[:nd, 3], basis_function=basis)
        call = lambda i: ctx.logl(A[i], Bc[i], C, Dd, tL, yL - theta[i, 5], theta[i, 4] * eL ** 2)
        ctx.set_option("tp_scan_tol", 1e30); call(1); ts = []
        for _ in range(5):
            tt = time.perf_counter(); call(1); ts.append(time.perf_counter() - tt)
        fast = min(ts)
        rows = {0: [], 1: []}
        refs, rsts = O.logl_batch(A, Bc, C, Dd, tL, yL, eL ** 2, theta[:nd, 5].copy(), theta[:nd, 4].copy(), nthreads=16, return_status=True)
        for i in range(nd):
            if rsts[i] or not np.isfinite(refs[i]): continue
            vw = refs[i]                                                                                   # reference: the ORACLE (the walk is not one: tools/tp_walk_accuracy.py)
            ctx.set_option("tp_scan_tol", 1e30); vs = call(i)                                              # the scan alone
            ctx.set_option("tp_scan_tol", None); vp = call(i)                                             # the product path
            es, ep = abs(vs - vw) / abs(vw), abs(vp - vw) / abs(vw)
            needs = []
            for chk, lad in ((3, ladder), (0, ladder_raw)):
                ctx.set_option("tp_check", chk)
                need = float("inf")
                for tol in lad:
                    ctx.set_option("tp_scan_tol", tol); call(i); tt = time.perf_counter(); call(i)
                    if time.perf_counter() - tt < 1.8 * fast: need = tol; break
                needs.append(need)
            rows[0].append((es, needs[0], ep, needs[1]))
            ctx.set_option("tp_scan_tol", None); ctx.set_option("tp_check", 0)
        allrows[0] += rows[0]
        es = np.array([r[0] for r in rows[0]]); ep = np.array([r[2] for r in rows[0]])
        print(f"{basis}-{nc} N={NL}: {len(es)} draws; scan alone off by more than 1e-8 on {(es > 1e-8).sum()} (max {es.max():.1e}); product path max {ep.max():.1e}", flush=True)
r = np.array(allrows[0])
print(f"# an estimate of log L's relative error from it (distance x sqrt(N) / |log L|; tp_check = 3); all {len(r)} draws:")
for tol in ladder:
    acc = r[:, 1] <= tol
    print(f"     threshold {tol:g}: {int(acc.sum())} accepted ({100 * (1 - acc.mean()):.1f} % repaired), worst error among the accepted {r[acc, 0].max() if acc.any() else 0:.1e}; "
          f"draws off by more than 1e-8: {int((r[acc, 0] > 1e-8).sum())} accepted, {int((r[~acc, 0] > 1e-8).sum())} rejected", flush=True)
print("# the distance on the innovation scale itself (the product's measure, tp_check = 0):")
for tol in ladder_raw:
    acc = r[:, 3] <= tol
    print(f"     threshold {tol:g}: {int(acc.sum())} accepted ({100 * (1 - acc.mean()):.1f} % repaired), worst error among the accepted {r[acc, 0].max() if acc.any() else 0:.1e}; "
          f"draws off by more than 1e-8: {int((r[acc, 0] > 1e-8).sum())} accepted", flush=True)
print("# accepted if EITHER is below its threshold (estimate | raw distance): share repaired, worst error among the accepted")
for t1 in (1e-9, 3e-9, 1e-8, 3e-8):
    print("     " + " | ".join(f"{t1:g}, {t2:g}: {100 * (1 - ((r[:, 1] <= t1) | (r[:, 3] <= t2)).mean()):.1f} % {r[(r[:, 1] <= t1) | (r[:, 3] <= t2), 0].max():.1e}" for t2 in (1e-7, 1e-6, 3e-6, 1e-5)), flush=True)
print("# accepted only if BOTH are below their thresholds")
for t1 in (1e-8, 3e-8, 1e-7, 1e-6):
    print("     " + " | ".join(f"{t1:g}, {t2:g}: {100 * (1 - ((r[:, 1] <= t1) & (r[:, 3] <= t2)).mean()):.1f} % {r[(r[:, 1] <= t1) & (r[:, 3] <= t2), 0].max():.1e}" for t2 in (1e-5, 1e-4, 1e-3, 1e-2)), flush=True)
