"""Numerical prototype (numpy, never shipped) of the windowed / blocked form of the celerite recurrence that the block kernel
(csrc/celerite_block.hip) runs: K time steps per window, the past enters through the R x R state T only via GEMMs,
the window's own K x K covariance block is eliminated by a small dense LDL^T.  Compared with the oracle's sequential restatement
of src/celerite_solver.jl:12-158 on the benchmark series.  Usage: python tools/block_proto.py [N] [K]"""
import sys
from pathlib import Path
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import oracle as O  # noqa: E402


def block_logl(a, b, c, d, t, y, s2, K=16):
    J = len(a); R = 2 * J + 1; N = len(t)
    al = np.zeros(R); be = np.zeros(R); cc = np.zeros(R); dd = np.zeros(R); sinrow = np.zeros(R, bool)
    al[0:2 * J:2] = a; be[0:2 * J:2] = b; al[1:2 * J:2] = a; be[1:2 * J:2] = -b
    cc[0:2 * J:2] = c; cc[1:2 * J:2] = c; dd[0:2 * J:2] = d; dd[1:2 * J:2] = d; sinrow[1:2 * J:2] = True
    T = np.zeros((R, R)); logdet = 0.0; quad = 0.0; Dall = []
    suma = a.sum()
    for m in range(0, N, K):
        n1 = min(N, m + K); k = n1 - m
        tt = t[m:n1]
        co = np.cos(np.outer(tt, dd)); si = np.sin(np.outer(tt, dd))          # [k][R]
        v = np.where(sinrow, si, co); x = np.where(sinrow, co, si)            # u = al v + be x
        v[:, R - 1] = y[m:n1]; x[:, R - 1] = 0.0
        tprev = t[m - 1] if m > 0 else t[0]
        C = np.exp(-np.outer(tt - tprev, cc))                                  # cumulative decay from the window base
        Cend = np.exp(-np.outer(tt[-1] - tt, cc))                              # step -> window end
        U = (al * v + be * x) * C                                              # scaled u, [k][R]
        M = U @ T                                                              # rows: steps; = (T u~)^T
        G = M @ U.T                                                            # Gram
        tau = np.abs(tt[:, None] - tt[None, :])
        A = np.zeros((k, k))
        # (cos, sin)(d tau) by angle addition from the same rounded (cos, sin)(d t) that U and v are made of — NOT np.cos(d * tau):
        # A must be consistent with the Gram block it is subtracted from (round 3; tools/explain_outliers.py)
        cj = np.cos(np.outer(tt, d)); sj = np.sin(np.outer(tt, d))
        later = tt[:, None] >= tt[None, :]
        for j in range(J):
            cd = np.outer(cj[:, j], cj[:, j]) + np.outer(sj[:, j], sj[:, j])
            sd = np.outer(sj[:, j], cj[:, j]) - np.outer(cj[:, j], sj[:, j])
            A += np.exp(-c[j] * tau) * (a[j] * cd + b[j] * np.where(later, sd, -sd))
        A[np.diag_indices(k)] = suma + s2[m:n1]
        S = A - G
        X = v * Cend - M * C[-1]                                               # [k][R]
        # LDL^T of S (no pivoting), L unit lower
        L = np.eye(k); D = np.zeros(k); W = S.copy()
        for p in range(k):
            D[p] = W[p, p]
            L[p + 1:, p] = W[p + 1:, p] / D[p]
            W[p + 1:, p + 1:] -= np.outer(L[p + 1:, p], W[p, p + 1:])
        Yt = np.linalg.solve(L, X)                                             # Yhat^T = L^-1 X^T
        T = T * np.outer(C[-1], C[-1]) + (Yt.T / D) @ Yt
        logdet += np.sum(np.log(np.abs(D))); quad += np.sum(Yt[:, R - 1] ** 2 / D)
        Dall.append(D)
    return -0.5 * logdet - 0.5 * N * np.log(2 * np.pi) - 0.5 * quad, np.concatenate(Dall)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    t, y, yerr = O.synthetic_series(N)
    for basis in ("SHO", "DRWCelerite"):
        th = O.synthetic_theta(8, t, y)
        A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, basis)
        worst = 0.0
        for i in range(8):
            ref = O.logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2)
            got, D = block_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2, K)
            rel = abs(got - ref) / abs(ref)
            worst = max(worst, rel)
            print(f"{basis} draw {i}: oracle {ref:.10f} block {got:.10f} rel {rel:.2e} minD {D.min():.3e}")
        print(f"{basis}: worst rel {worst:.2e}")
