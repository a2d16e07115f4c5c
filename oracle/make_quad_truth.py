#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generates tests/golden/quad_truth.npz — ill-conditioned prior draws of the bench model (SHO-20) with log L evaluated
in __float128 (oracle/celerite_oracle_q.c), the truth against which the fp64 oracle and every GPU kernel family are held
(tests/test_oracle.py::test_fp64_oracle_vs_quad_truth, tests/test_gpu_parity.py::test_ill_conditioned_draws_vs_quad_truth).

The draws are the ones tools/block_accuracy_scan.py uses (same seeds): PRIOR draws, a tenth of them with nu scaled down by 10 .. 1000 on
purpose; kept are all draws with ratio = nu min(sigma2) / sum(a) < 1e-7 that the fp64 oracle calls positive definite, and 40 draws of
each of the bins [1e-7, 1e-6), [1e-6, 1e-5).  Three series: N = 150, 1000 (seeded) and the bench series N = 1e4 (O.synthetic_series).
CPU only; ~10 minutes on 8 cores (libquadmath is software arithmetic: ~3 ms per time step at 40 rows).

usage: python oracle/make_quad_truth.py [threads]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = {}
for N, B, seed in ((150, 6000, 1), (1000, 3000, 2), (10000, 1024, 3)):
    if N == 10000:
        t, y, yerr = O.synthetic_series(N, seed=1234)
    else:
        rng = np.random.default_rng(seed)
        t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); yerr = rng.uniform(0.01, 0.05, N)
    th = O.synthetic_theta(B, t, y, seed=seed)
    rng = np.random.default_rng(seed + 10)
    k = rng.choice(B, B // 10, replace=False); th[k, 4] *= 10.0 ** (-rng.uniform(1, 3, len(k)))
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, "SHO")
    ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, yerr ** 2, mu, nu, nthreads=nthreads, return_status=True)
    ratio = nu * np.min(yerr ** 2) / A.sum(axis=1)
    ok = (rst == 0) & np.isfinite(ref)
    pick = list(np.flatnonzero(ok & (ratio < 1e-7)))
    rs = np.random.default_rng(seed + 20)
    for lo, hi in ((1e-7, 1e-6), (1e-6, 1e-5)):
        cand = np.flatnonzero(ok & (ratio >= lo) & (ratio < hi))
        pick += list(rs.choice(cand, min(40, len(cand)), replace=False))
    pick = np.array(sorted(pick))
    t0 = time.time()
    truth, dmin = O.logl_quad_batch(A[pick], Bc[pick], C, Dd, t, y, yerr ** 2, mu[pick], nu[pick], nthreads=nthreads, return_dmin=True)
    e = np.abs(ref[pick] - truth) / np.abs(truth)
    print(f"N = {N}: {len(pick)} draws, {time.time() - t0:.0f} s; fp64 oracle vs truth: max {e.max():.2e}, median {np.median(e):.2e}; min D over the draws {dmin.min():.3e}",
          flush=True)
    tag = f"n{N}"
    if N != 10000:
        out[f"{tag}_t"] = t; out[f"{tag}_y"] = y; out[f"{tag}_yerr"] = yerr
    out[f"{tag}_A"] = A[pick]; out[f"{tag}_Bc"] = Bc[pick]; out[f"{tag}_C"] = C; out[f"{tag}_Dd"] = Dd
    out[f"{tag}_mu"] = mu[pick]; out[f"{tag}_nu"] = nu[pick]; out[f"{tag}_ratio"] = ratio[pick]
    out[f"{tag}_truth"] = truth; out[f"{tag}_dmin"] = dmin; out[f"{tag}_oracle_fp64"] = ref[pick]
np.savez_compressed(ROOT / "tests" / "golden" / "quad_truth.npz", **out)
print("wrote tests/golden/quad_truth.npz")
