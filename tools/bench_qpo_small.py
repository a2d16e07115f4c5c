#!/usr/bin/env python3
"""GPU: the QPO model at a nested sampler's batch sizes (SHO-20 continuum + 1 sampled QPO term, J = 21, N = 1e4): the windowed
kernel with per-draw rows (default up to 512 draws) against the latency layout's mixed mode ("no_block")."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
N, J = 10_000, 20
t, y, yerr = bench.synth_series(N)
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
res = {"workload": f"N={N}, SHO-{J} + 1 QPO term per draw (J=21), host-pointer entry (PCIe included)", "rows": []}
for B in (64, 256, 400, 512):
    th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
    rng = np.random.default_rng(3)
    f0 = np.exp(rng.uniform(np.log(1e-2), np.log(1.0), B)); Q = rng.uniform(2, 20, B); S0 = rng.uniform(0.01, 0.1, B)
    qa = np.array([pj.convert_feature(pj.QPO(S0[i], f0[i], Q[i])) for i in range(B)])
    A2 = np.concatenate([A, 2 * qa[:, :1]], axis=1); B2 = np.concatenate([Bc, 2 * qa[:, 1:2]], axis=1)
    C2 = np.concatenate([np.broadcast_to(C, (B, J)), qa[:, 2:3]], axis=1); D2 = np.concatenate([np.broadcast_to(Dd, (B, J)), qa[:, 3:4]], axis=1)
    row = {"B": B}
    outs = {}
    for mode in ("default", "no_block"):
        ctx.set_option("no_block", mode == "no_block")
        ds.logl_batch(A2, B2, C2, D2, mu=th[:, 5], nu=th[:, 4])
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = ds.logl_batch(A2, B2, C2, D2, mu=th[:, 5], nu=th[:, 4]); ts.append(time.perf_counter() - t0)
        row[mode] = {"kernel": name(), "ms_per_call": round(1e3 * float(np.median(ts)), 3)}
        outs[mode] = out
    ctx.set_option("no_block", False)
    ok = np.isfinite(outs["default"]) & np.isfinite(outs["no_block"])
    row["max_rel_diff"] = float(np.max(np.abs(outs["default"][ok] - outs["no_block"][ok]) / np.abs(outs["no_block"][ok])))
    if B == 64:
        from oracle import oracle as O
        ref = O.logl_batch(A2[:16], B2[:16], C2[:16], D2[:16], t, y, yerr ** 2, th[:16, 5].copy(), th[:16, 4].copy(), nthreads=16)
        row["max_rel_err_vs_oracle"] = float(np.nanmax(np.abs(outs["default"][:16] - ref) / np.abs(ref)))
    res["rows"].append(row)
    print(row, flush=True)
print(json.dumps(res))
