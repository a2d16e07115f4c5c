#!/usr/bin/env python3
"""GPU: one scalar evaluation (host entry, PCIe included), serial-chain kernels ("no_tp") against the time-parallel family with its boundary phase as a scan (forced: scan_config
"tp", tp_scan 1) and as the walk (tp_scan 0), by series length and rows: where the dispatch thresholds of capi.hip tp_dispatch belong."""
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
    for _ in range(7):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for N in (512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 10000, 16384, 65536):
    t, y, yerr = bench.synth_series(N)
    th, f_min, f_max = bench.synth_theta(4, t, y, seed=99)
    line = f"N={N}:"
    wide = len(sys.argv) > 1 and sys.argv[1] == "wide"          # 49 .. 64 state rows (tp_combine_lean_kernel)
    for basis, nc in ((("SHO", 28), ("DRWCelerite", 20), ("SHO", 32)) if wide else (("SHO", 4), ("SHO", 8), ("SHO", 12), ("SHO", 16), ("SHO", 20), ("SHO", 24))):
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
        a, b, mu, nu = A[1], Bc[1], th[1, 5], th[1, 4]
        f = lambda: ctx.logl(a, b, C, Dd, t, y - mu, nu * yerr ** 2)
        ctx.set_option("no_tp", True); ser = timed(f); ks = name(); ctx.set_option("no_tp", False)
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 1)
        try: sc = timed(f)
        except Exception: sc = float("nan")
        ctx.set_option("tp_scan", 0)
        try: wk = timed(f)
        except Exception: wk = float("nan")
        ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1)
        auto = timed(f); ka = name()
        line += f"  {len(C) * 2 - int((np.asarray(Dd) == 0).sum())}r: {ks} {ser:.3f} | scan {sc:.3f} | walk {wk:.3f} | auto[{ka}] {auto:.3f}"
    print(line, flush=True)
