#!/usr/bin/env python3
"""GPU, experiment build (tools/tp_combine_stamps.sh): where a combination of the time-parallel scan spends its time — s_memtime stamps (ticks; shares are what counts) of the last
target's workgroup in the LAST launch of a combination kernel (the last level of the scan at 128 segments: a prefix-mode combination), thread 0's view; for tp_combine_kernel also
the inside of the first elimination round.  usage (GPU box): PIORAN_HIP_LIB=$PWD/tools/experiments/libpioran_stamp.so python tools/tp_combine_stamps.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init()
import bench, pioran_jl_amd as pj
L = pj._lib.lib()
L.pioran_tp_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
ctx = pj.Context(0)
N = 10000
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(8, t, y, seed=4321)
names = ["barrier after zeroing", "operands -> LDS", "W, z, copies", "elimination", "scale + permute", "vectors + products", "C, outputs"]
for basis, nc in (("SHO", 20), ("DRWCelerite", 20), ("SHO", 12), ("SHO", 8)):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, nc, th[:, 3], basis_function=basis)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 1); ctx.set_option("tp_segments", 128)
    acc = np.zeros(7); inner = np.zeros(4); n = 0
    for rep in range(6):
        ds.logl_batch(A[:1], Bc[:1], C, Dd, mu=th[:1, 5].copy(), nu=th[:1, 4].copy())
        buf = (ctypes.c_ulonglong * 32)()
        assert L.pioran_tp_read_stamps(buf) == 0
        s = np.array(buf[:8], dtype=np.float64); r = np.array(buf[10:14], dtype=np.float64)
        if rep == 0: continue
        acc += np.diff(s); inner += np.array([r[0] - s[3], r[1] - r[0], r[2] - r[1], r[3] - r[2]]); n += 1
    ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_segments", 0)
    tk = acc / n
    print(f"{basis}-{nc}: {tk.sum():.0f} ticks | " + " | ".join(f"{nm} {v:.0f} ({v / tk.sum():.0%})" for nm, v in zip(names, tk)), flush=True)
    if basis == "SHO" and nc > 8:
        iv = inner / n
        print(f"      first elimination round (four pivots) of {int(np.ceil(2 * nc / 4))}: pivot search {iv[0]:.0f} | multipliers between the pivot rows, to LDS {iv[1]:.0f} | rank-4 update of this wavefront's column tile {iv[2]:.0f} | "
              f"waiting at the barrier {iv[3]:.0f}", flush=True)
