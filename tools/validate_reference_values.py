#!/usr/bin/env python3
"""Every log-likelihood the REFERENCE itself computed and stored (five runs, 34 271 values: tests/golden/ultranest_points.npz,
ultranest_example_runs.npz, turing_chain.npz) through each kernel family that takes the shape, forced: the automatic
choice, the tile kernel (windowed form, one draw per wavefront), the step-by-step throughput layout, the small-batch windowed kernel (chunks of
256 draws) and the time-parallel family (chunks of 64 draws, 40 state rows: four wavefronts per segment).  Per family: the kernel that ran,
median / 99.9 % / maximum relative deviation from the reference's value.  GPU box: python tools/validate_reference_values.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import pioran_jl_amd as pj
G = "tests/golden/"
ctx = pj.Context(0)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()


def runs():
    un = np.load(G + "ultranest_points.npz")
    t, y, yerr, P, ref = un["t"], un["y"], un["yerr"], un["params"], un["logl"]
    f = (1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, P[:, :3], *f, 20, P[:, 3], is_integrated_power=False)
    cs = P[:, 6:7]
    yield "ultranest run (docs/src/data/inference; N = 242, sampled shift: per-draw series)", t, y, yerr ** 2, A, Bc, C, Dd, P[:, 5], P[:, 4], np.log(y[None, :] - cs), yerr[None, :] ** 2 / (y[None, :] - cs) ** 2, ref
    ex = np.load(G + "ultranest_example_runs.npz")
    for nm in ("simu_single", "simu_double", "simu_periodic"):
        t, y, yerr, P, ref = (ex[f"{nm}_{k}"] for k in ("t", "y", "yerr", "params", "logl"))
        f = (1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2)
        if nm == "simu_double":
            A, Bc, C, Dd = pj.approx_batch(pj.DoubleBendingPowerLaw, P[:, :5], *f, 20, P[:, 5])
            k = 6
        else:
            A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, P[:, :3], *f, 20, P[:, 3])
            k = 4
        if nm == "simu_periodic":
            amp, ph, T0 = P[:, 6:7], P[:, 7:8], P[:, 8:9]
            Y = y[None, :] - amp * np.sin(2 * np.pi * t[None, :] / T0 + ph)
            yield f"examples/ultranest {nm} (N = {len(t)}, CustomMean: per-draw series)", t, y, yerr ** 2, A, Bc, C, Dd, P[:, 5], P[:, 4], Y, np.broadcast_to(yerr[None, :] ** 2, Y.shape).copy(), ref
        else:
            yield f"examples/ultranest {nm} (N = {len(t)})", t, np.log(y), yerr ** 2 / y ** 2, A, Bc, C, Dd, P[:, k + 1], P[:, k], None, None, ref
    tc = np.load(G + "turing_chain.npz")
    t, y, yerr, P, ref = tc["t"], tc["y"], tc["yerr"], tc["params"], tc["logl"]
    f = (1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, P[:, :3], *f, 20, P[:, 3], is_integrated_power=False)
    cs = P[:, 6:7]
    yield "NUTS chains (docs/src/data/subset_simu_single.h5; N = 250, sampled shift)", t, y, yerr ** 2, A, Bc, C, Dd, P[:, 5], P[:, 4], np.log(y[None, :] - cs), yerr[None, :] ** 2 / (y[None, :] - cs) ** 2, ref


FAMILIES = (("automatic", {}, 0), ("tile (forced)", {"scan_config": "tile"}, 0), ("step-by-step layout (no_tile, no_block)", {"no_tile": True, "no_block": True}, 0),
            ("small-batch windowed kernel, 256 draws per call", {"no_tile": True}, 256), ("time-parallel (forced), 64 draws per call", {"scan_config": "tp"}, 64))
total = 0
for title, t, y, s2, A, Bc, C, Dd, mu, nu, Y, S2, ref in runs():
    ds = pj.Dataset(t, y, s2, ctx)
    B = len(ref); total += B
    print(f"{title}: {B} values", flush=True)
    for fam, opts, chunk in FAMILIES:
        for k, v in opts.items(): ctx.set_option(k, v)
        try:
            got = np.empty(B); kern = set()
            step = chunk or B
            for b0 in range(0, B, step):
                sl = slice(b0, min(B, b0 + step))
                got[sl] = ds.logl_batch(A[sl], Bc[sl], C, Dd, mu=mu[sl], nu=nu[sl], Y=None if Y is None else Y[sl], S2=None if S2 is None else S2[sl])
                kern.add(name())
        finally:
            for k in opts: ctx.set_option(k, None)
        e = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        print(f"    {fam:52s} [{', '.join(sorted(kern))}]: median {np.median(e):.1e}  99.9 % {np.quantile(e, 0.999):.1e}  max {e.max():.1e}", flush=True)
    ds.close()
print(f"{total} reference-computed values")
