#!/usr/bin/env python3
"""GPU: mid-size batches (512 < B <= 4096: more draws than the windowed kernel likes, fewer than fill the chip with the
many-draws-per-wavefront shapes).  ms per resident launch at N = 1e4 for the default choice, for throughput shapes with fewer
draws per wavefront (selected by name) and for the windowed kernel forced.  usage: python tools/sweep_midbatch.py [J ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj

N = int(os.environ.get("N", 10_000))
JS = [int(a) for a in sys.argv[1:]] or [5, 10, 15, 20, 23]
BS = [256, 512, 768, 1024, 1536, 2048, 3072, 4096]
ALTS = ["rpl1_cbr1_nsrc16", "rpl2_cbr2_nsrc8", "rpl3_cbr2_nsrc7_p", "rpl3_cbr4_nsrc4", "rpl4_cbr4_nsrc4", "block"]
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
th, f_min, f_max = bench.synth_theta(max(BS), t, y, seed=4321)
name = lambda: pj._lib.lib().pioran_celerite_config_name(0).decode()


def med_ms(f, reps=4):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for basis in ("SHO", "DRWCelerite"):
    for J in JS:
        if basis == "DRWCelerite" and J not in (10, 20):
            continue
        A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3], basis_function=basis)
        real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
        R = int(2 * len(C) - real.sum())
        ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real.astype(np.int32))
        d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
        dout = torch.empty(max(BS), dtype=torch.float64, device=dev); dst = torch.zeros(max(BS), dtype=torch.int32, device=dev)
        print(f"# {basis}-{J}: {R} rows; columns: B default(ms, config) | " + " | ".join(ALTS), flush=True)
        for B in BS:
            go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), dst.data_ptr())
            ctx.set_option("scan_config", None)
            base = med_ms(go); fam = pj._lib.lib().pioran_celerite_config_name(-1).decode(); cfg0 = name() if fam == "scan" else f"({fam})"
            ref = dout[:B].clone(); valid = dst[:B] == 0      # (draws without a positive definite covariance, status 1, are not compared)
            cells = []
            for alt in ALTS:
                ctx.set_option("scan_config", alt)
                if alt != "block": ctx.set_option("no_wide", True)
                ms = med_ms(go)
                ran = name() if alt != "block" else "block"
                ok = bool(torch.allclose(dout[:B][valid], ref[valid], rtol=1e-9, atol=0, equal_nan=True))
                cells.append(f"{ms:7.3f}{'' if ran == alt else '*'}{'' if ok else '!'}")
                ctx.set_option("no_wide", None)
            ctx.set_option("scan_config", None)
            print(f"{B:5d} {base:7.3f} {cfg0:22s} | " + " | ".join(cells) + f"   best {B / min([base] + [float(c.rstrip('*!')) for c in cells]) :8.0f} evals/ms", flush=True)
        ds.close()
print("(* = the named shape does not hold this row count: another one ran; ! = differs from the default's values by more than 1e-9 on the draws with status 0)")
