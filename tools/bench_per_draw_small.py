#!/usr/bin/env python3
"""GPU: small batches whose (c, d) ALL differ per draw (free Celerite terms / CARMA kernels under a sampler, src/CARMA.jl:98-143):
host-pointer entry, N = 1e4, J = 20 two-row terms.  ms per call by batch size, kernel family, agreement with the oracle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
N, J = 10_000, int(os.environ.get("J", 20))
t, y, yerr = bench.synth_series(N)
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
rows = []
for B in (1, 16, 64, 256):
    th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
    rng = np.random.default_rng(5)
    C2 = np.broadcast_to(C, (B, J)) * rng.uniform(0.9, 1.1, (B, J)); D2 = np.broadcast_to(Dd, (B, J)) * rng.uniform(0.9, 1.1, (B, J))
    ds.logl_batch(A, Bc, C2, D2, mu=th[:, 5], nu=th[:, 4])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = ds.logl_batch(A, Bc, C2, D2, mu=th[:, 5], nu=th[:, 4]); ts.append(time.perf_counter() - t0)
    row = {"B": B, "ms_per_call": round(1e3 * float(np.median(ts)), 3), "kernel": name()}
    if B <= 16:
        from oracle import oracle as O
        ref, rst = O.logl_batch(A, Bc, C2, D2, t, y, yerr ** 2, th[:, 5].copy(), th[:, 4].copy(), nthreads=16, return_status=True)
        ok = rst == 0     # (the random perturbation of (c, d) leaves a few draws without a positive definite covariance: status 1 in both)
        row["max_rel_err_vs_oracle_valid_draws"] = float(np.nanmax(np.abs(out[ok] - ref[ok]) / np.abs(ref[ok])))
        row["valid_draws"] = int(ok.sum())
    rows.append(row); print(row, flush=True)
print(json.dumps({"workload": f"N={N}, SHO-{J} with per-draw (c, d) in every term, host-pointer entry", "rows": rows}))
