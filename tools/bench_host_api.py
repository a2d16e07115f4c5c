#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batch entry (pioran_celerite_logl_batch): per call H2D of A, Bc, mu,
nu (1.3 MB at B=4096, J=20), launch, D2H of logL/status, plus the host-side approx_batch that produces (A, Bc)
from theta.  Reported next to the HBM-resident `value` of bench.py (never instead of it)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
N, B, J = 10_000, 4096, 20
t, y, yerr = bench.synth_series(N)
th, f_min, f_max = bench.synth_theta(B, t, y, seed=4321)
ctx = pj.Context(0); ds = pj.Dataset(t, y, yerr ** 2, ctx)
def prep(): return pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
A, Bc, C, Dd = prep()
ds.logl_batch(A, Bc, C, Dd, mu=th[:, 5], nu=th[:, 4])
tp, tc = [], []
for _ in range(5):
    t0 = time.perf_counter(); A, Bc, C, Dd = prep(); t1 = time.perf_counter()
    out = ds.logl_batch(A, Bc, C, Dd, mu=th[:, 5], nu=th[:, 4]); t2 = time.perf_counter()
    tp.append(t1 - t0); tc.append(t2 - t1)
tp, tc = float(np.median(tp)), float(np.median(tc))
# theta-only entry: approx on the device
ds.logpdf_theta(pj.SingleBendingPowerLaw, th[:, :3], th[:, 3], f_min, f_max, J, mu=th[:, 5], nu=th[:, 4])
tt = []
for _ in range(5):
    t0 = time.perf_counter(); o2 = ds.logpdf_theta(pj.SingleBendingPowerLaw, th[:, :3], th[:, 3], f_min, f_max, J, mu=th[:, 5], nu=th[:, 4]); tt.append(time.perf_counter() - t0)
tt = float(np.median(tt))
dev_vs_host = float(np.nanmax(np.abs(o2 - out) / np.abs(out)))
# single-evaluation latency of the scalar drop-in (config 2: B = 1, N = 1e4)
R = pj.approx(pj.SingleBendingPowerLaw(0.82, 0.01, 3.3), f_min, f_max, J, 1.0)
pj.logl(R.a, R.b, R.c, R.d, t, y, yerr ** 2, ctx=ctx)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); pj.logl(R.a, R.b, R.c, R.d, t, y, yerr ** 2, ctx=ctx); ts.append(time.perf_counter() - t0)
t1k = []
for _ in range(5):
    t0 = time.perf_counter(); pj.logl(R.a, R.b, R.c, R.d, t[:1000], y[:1000], yerr[:1000] ** 2, ctx=ctx); t1k.append(time.perf_counter() - t0)
print(json.dumps({"workload": f"N={N}, SHO-{J}, B={B}", "host_api_call_ms": tc * 1e3, "host_api_evals_per_s": B / tc,
                  "approx_batch_ms": tp * 1e3, "end_to_end_evals_per_s_incl_approx": B / (tc + tp),
                  "theta_only_call_ms": tt * 1e3, "theta_only_evals_per_s": B / tt, "theta_only_vs_host_approx_max_rel": dev_vs_host,
                  "scalar_logl_ms_N1e4": float(np.median(ts)) * 1e3, "scalar_logl_ms_N1e3": float(np.median(t1k)) * 1e3}))
