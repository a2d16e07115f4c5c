#!/usr/bin/env python3
"""CPU (numpy): which stage of the windowed form carries its error on ill-conditioned draws?  The windowed algorithm of celerite_block.hip /
celerite_tile.hip restated with a switch per stage between fp64 and x87 extended precision (64-bit significand: 2000 times finer — "exact"
for one stage at a time), on the worst draws of tests/golden/quad_truth.npz, against the __float128 truth.
usage: python tools/window_precision_study.py [N: 150 | 1000] [draws]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
LD = np.longdouble


def block_logl(a, b, c, d, t, y, s2, hi=(), K=16):
    """hi: stages in extended precision, of 'table' (trigonometric / decay factors), 'A', 'MG' (M' and the Gram block), 'sub' (Sigma = A - G),
    'ldl', 'X' (X', Y^'), 'upd' (the update and the storage of T)"""
    P = lambda s: LD if s in hi else np.float64
    J = len(a); R = 2 * J + 1; N = len(t)
    al = np.zeros(R); be = np.zeros(R); cc = np.zeros(R); dd = np.zeros(R); sinrow = np.zeros(R, bool)
    al[0:2 * J:2] = a; be[0:2 * J:2] = b; al[1:2 * J:2] = a; be[1:2 * J:2] = -b
    cc[0:2 * J:2] = c; cc[1:2 * J:2] = c; dd[0:2 * J:2] = d; dd[1:2 * J:2] = d; sinrow[1:2 * J:2] = True
    T = np.zeros((R, R), dtype=P('upd')); logdet = LD(0); quad = LD(0)
    suma = a.sum()
    for m in range(0, N, K):
        n1 = min(N, m + K); k = n1 - m
        pt = P('table')
        tt = t[m:n1].astype(pt)
        # the phases d * t are rounded to fp64 in every fp64 path (the kernels' tables): kept that way unless 'table' is extended
        ph = np.outer(tt, dd.astype(pt)) if 'table' in hi else np.outer(t[m:n1], dd).astype(pt)
        co = np.cos(ph); si = np.sin(ph)
        v = np.where(sinrow, si, co); x = np.where(sinrow, co, si)
        v[:, R - 1] = y[m:n1]; x[:, R - 1] = 0.0
        tprev = t[m - 1] if m > 0 else t[0]
        C = np.exp(-np.outer(tt - pt(tprev), cc.astype(pt)))
        Cend = np.exp(-np.outer(tt[-1] - tt, cc.astype(pt)))
        U = ((al * v + be * x) * C).astype(np.float64 if 'table' not in hi else pt)
        pm = P('MG')
        M = U.astype(pm) @ T.astype(pm)
        G = M @ U.astype(pm).T
        pa = P('A')
        tau = np.abs(tt[:, None] - tt[None, :]).astype(pa)
        A = np.zeros((k, k), dtype=pa)
        cj = co[:, 0:2 * J:2].astype(pa); sj = si[:, 0:2 * J:2].astype(pa)
        later = tt[:, None] >= tt[None, :]
        for j in range(J):
            cd = np.outer(cj[:, j], cj[:, j]) + np.outer(sj[:, j], sj[:, j])
            sd = np.outer(sj[:, j], cj[:, j]) - np.outer(cj[:, j], sj[:, j])
            A += np.exp(-pa(c[j]) * tau) * (pa(a[j]) * cd + pa(b[j]) * np.where(later, sd, -sd))
        A[np.diag_indices(k)] = pa(suma) + s2[m:n1].astype(pa)
        ps = P('sub')
        S = (A.astype(ps) - G.astype(ps))
        px = P('X')
        X = (v * Cend).astype(px) - M.astype(px) * C[-1].astype(px)
        pl = P('ldl')
        L = np.eye(k, dtype=pl); D = np.zeros(k, dtype=pl); W = S.astype(pl).copy()
        for p in range(k):
            D[p] = W[p, p]
            L[p + 1:, p] = W[p + 1:, p] / D[p]
            W[p + 1:, p + 1:] -= np.outer(L[p + 1:, p], W[p, p + 1:])
        # Y^' = L^-1 X' by forward substitution in the stage's precision
        Yt = X.astype(px).copy(); Lx = L.astype(px)
        for p in range(k):
            Yt[p + 1:] -= np.outer(Lx[p + 1:, p], Yt[p])
        pu = P('upd')
        T = T * np.outer(C[-1], C[-1]).astype(pu) + (Yt.T.astype(pu) / D.astype(pu)) @ Yt.astype(pu)
        logdet += np.sum(np.log(np.abs(D.astype(LD)))); quad += np.sum(Yt[:, R - 1].astype(LD) ** 2 / D.astype(LD))
    return float(-LD(0.5) * logdet - LD(0.5) * N * np.log(2 * LD(np.pi)) - LD(0.5) * quad)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    nd = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    q = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / "quad_truth.npz")
    tag = f"n{N}"
    t, y, yerr = q[f"{tag}_t"], q[f"{tag}_y"], q[f"{tag}_yerr"]
    A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
    truth, ratio, orc = q[f"{tag}_truth"], q[f"{tag}_ratio"], q[f"{tag}_oracle_fp64"]
    base = np.array([block_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2) for i in range(len(truth))]) if nd >= len(truth) else None
    sel = np.flatnonzero((ratio < 1e-7))
    e0 = np.array([abs(block_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2) - truth[i]) / abs(truth[i]) for i in sel])
    worst = sel[np.argsort(-e0)[:nd]]
    variants = [(), ('table',), ('A',), ('MG',), ('sub',), ('A', 'MG', 'sub'), ('ldl',), ('X',), ('upd',), ('ldl', 'X', 'upd'), ('A', 'MG', 'sub', 'ldl', 'X', 'upd'),
                ('table', 'A', 'MG', 'sub', 'ldl', 'X', 'upd')]
    print(f"N = {N}: the {nd} draws with ratio < 1e-7 on which the fp64 windowed form is furthest from the truth; relative deviation from the truth per variant")
    print("draw ratio     oracle   " + "  ".join("+".join(v) if v else "fp64" for v in variants))
    for i in worst:
        row = [abs(block_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2, hi=v) - truth[i]) / abs(truth[i]) for v in variants]
        print(f"{i:4d} {ratio[i]:.1e}  {abs(orc[i] - truth[i]) / abs(truth[i]):.1e}  " + "  ".join(f"{e:.1e}" for e in row), flush=True)
