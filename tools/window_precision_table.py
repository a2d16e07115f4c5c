#!/usr/bin/env python3
"""CPU: the per-stage table VERDICT round 5 asked for — maximum and 90th percentile of the windowed form's relative deviation from the __float128 truth
over EVERY draw of tests/golden/quad_truth.npz below a conditioning ratio, with one stage at a time in x87 extended precision
(tools/window_precision_study.py's restatement).  usage: python tools/window_precision_table.py [N: 150 | 1000] [ratio bound, default 1e-8]"""
import sys
from multiprocessing import Pool
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tools.window_precision_study import block_logl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
bound = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
q = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / "quad_truth.npz")
tag = f"n{N}"
t, y, yerr = q[f"{tag}_t"], q[f"{tag}_y"], q[f"{tag}_yerr"]
A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
truth, ratio, orc = q[f"{tag}_truth"], q[f"{tag}_ratio"], q[f"{tag}_oracle_fp64"]
sel = np.flatnonzero(ratio < bound)
variants = [(), ('A',), ('MG',), ('sub',), ('A', 'MG', 'sub'), ('ldl',), ('X',), ('upd',), ('ldl', 'X', 'upd'), ('A', 'MG', 'sub', 'ldl', 'X', 'upd')]


def one(args):
    i, v = args
    return abs(block_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2, hi=v) - truth[i]) / abs(truth[i])


if __name__ == "__main__":
    print(f"# N = {N}, {len(sel)} draws with ratio < {bound:g}; relative deviation from the __float128 truth; fp64 oracle: max {np.max(np.abs(orc[sel] - truth[sel]) / np.abs(truth[sel])):.2e}, "
          f"90th percentile {np.percentile(np.abs(orc[sel] - truth[sel]) / np.abs(truth[sel]), 90):.2e}")
    print("# stage(s) in extended precision        max        90th pct   median")
    with Pool(8) as pool:
        for v in variants:
            e = np.array(pool.map(one, [(i, v) for i in sel]))
            print(f"{'+'.join(v) if v else 'none (fp64)':38s} {e.max():.2e}   {np.percentile(e, 90):.2e}   {np.median(e):.2e}", flush=True)
