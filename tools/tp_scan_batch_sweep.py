#!/usr/bin/env python3
"""GPU: a handful of draws (3 .. 64) of one long series: serial-chain kernels ("no_tp") against the time-parallel family with the boundary walk and with the boundary scan
at several segment counts — where the scan's dispatch (up to two draws in round 6's first cut) can go.  Resident inputs (Dataset.logl_batch), host entry."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench, pioran_jl_amd as pj
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
def timed(f):
    f(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
Ns = [int(a) for a in sys.argv[1:]] or [10000]
for N in Ns:
    t, y, yerr = bench.synth_series(N)
    th, f_min, f_max = bench.synth_theta(64, t, y, seed=99)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    for basis, nc in (("SHO", 4), ("SHO", 8), ("SHO", 12), ("SHO", 20), ("DRWCelerite", 20)):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
        for B in (3, 4, 6, 8, 12, 16, 32, 64):
            f = lambda: ds.logl_batch(A[:B], Bc[:B], C, Dd, mu=th[:B, 5].copy(), nu=th[:B, 4].copy())
            ctx.set_option("no_tp", True); ser = timed(f); ks = name(); ctx.set_option("no_tp", False)
            auto = timed(f); ka = name()
            line = f"{basis}-{nc} N={N} B={B}: {ks} {ser:.3f} | auto[{ka}] {auto:.3f}"
            ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 0)
            try: line += f" | walk {timed(f):.3f}"
            except Exception: line += " | walk -"
            ctx.set_option("tp_scan", 1)
            for sg in (0, 16, 32, 64, 128):
                ctx.set_option("tp_segments", sg)
                try: line += f" | scan/{sg} {timed(f):.3f}"
                except Exception: line += f" | scan/{sg} -"
            ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_segments", 0)
            print(line, flush=True)
