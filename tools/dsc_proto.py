"""Numerical prototype (numpy, never shipped) of the DEFERRED-SCALING form of the celerite recurrence
(src/celerite_solver.jl:52-97,136-141): inside a window that starts at step s the state is kept in the coordinates
    T~ = T ./ (Psi Psi'),   Psi_n = prod_{s < k <= n} phi_k   (element-wise, Psi_s = 1)
so that the Hadamard scaling by phi phi' (:78-79,85) disappears from the steps of the window:
    u~_n = Psi_n o u_n,  v~_n = v_n ./ Psi_n          (table entries: they depend on (c, d, t) only)
    r = T~ u~_n;  D_n = d_n - u~_n' r;  m~ = v~_n - r;  T~ += m~ m~' / D_n
and T = (P P') o T~ is applied once at the next window boundary (P = the window's total decay).  A window ends after K steps
or when the accumulated exponent max_j c_j (t_n - t_s) would pass L (so 1 / Psi stays far from overflow).
Compared with the oracle's sequential restatement on the benchmark series.  Usage: python tools/dsc_proto.py [N] [K] [L]"""
import sys
from pathlib import Path
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import oracle as O  # noqa: E402


def boundaries(t, cmax, K, L):
    """flag[n] = a window starts at step n (fixed grid n % K == 0, plus wherever the exponent since the last start passes L)."""
    N = len(t)
    flag = np.zeros(N, bool); start = np.zeros(N, np.int64)
    s = 0; E = 0.0
    for n in range(N):
        e = cmax * (t[n] - t[n - 1]) if n > 0 else 0.0
        if n % K == 0 or E + e > L:
            s = n; E = 0.0
        else:
            E += e
        flag[n] = s == n; start[n] = s
    return flag, start


def dsc_logl(a, b, c, d, t, y, s2, K=32, L=100.0, dtype=np.float64):
    J = len(a); R = 2 * J + 1; N = len(t)
    al = np.zeros(R); be = np.zeros(R); cc = np.zeros(R); dd = np.zeros(R); sinrow = np.zeros(R, bool)
    al[0:2 * J:2] = a; be[0:2 * J:2] = b; al[1:2 * J:2] = a; be[1:2 * J:2] = -b
    cc[0:2 * J:2] = c; cc[1:2 * J:2] = c; dd[0:2 * J:2] = d; dd[1:2 * J:2] = d; sinrow[1:2 * J:2] = True
    flag, start = boundaries(t, c.max(), K, L)
    T = np.zeros((R, R), dtype); logdet = 0.0; quad = 0.0
    suma = a.sum(); nwin = 0; maxE = 0.0
    m = np.zeros(R, dtype); w = np.zeros(R, dtype)
    for n in range(N):
        co = np.cos(dd * t[n]); si = np.sin(dd * t[n])
        v = np.where(sinrow, si, co); x = np.where(sinrow, co, si)
        v[R - 1] = 0.0; x[R - 1] = 0.0
        if flag[n]:
            sprev = start[n - 1] if n > 0 else 0
            P = np.exp(-cc * (t[n] - t[sprev])) if n > 0 else np.zeros(R)
            P[R - 1] = 1.0 if n > 0 else 0.0
            T = np.outer(P, P) * (T + np.outer(m, w))      # the pending rank-1 term, then back to true scale
            nwin += 1
        else:
            T = T + np.outer(m, w)
        E = cc * (t[n] - t[start[n]]); maxE = max(maxE, E.max())
        Psi = np.exp(-E)
        tv = Psi * v; tx = Psi * x; vv = v * np.exp(E)     # the three table entries of the row
        vv[R - 1] = y[n]
        u = al * tv + be * tx
        r = T @ u
        D = suma + s2[n] - u @ r
        m = vv - r
        w = m / D
        logdet += np.log(abs(D)); quad += m[R - 1] ** 2 / D
    return -0.5 * logdet - 0.5 * N * np.log(2 * np.pi) - 0.5 * quad, nwin, maxE


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    L = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    t, y, yerr = O.synthetic_series(N)
    for basis in ("SHO", "DRWCelerite"):
        th = O.synthetic_theta(8, t, y)
        A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, basis)
        worst = 0.0
        for i in range(8):
            ref = O.logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2)
            got, nwin, maxE = dsc_logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2, K, L)
            rel = abs(got - ref) / abs(ref)
            worst = max(worst, rel)
            print(f"{basis} draw {i}: oracle {ref:.10f} deferred {got:.10f} rel {rel:.2e} windows {nwin} (mean {N / nwin:.1f} steps) max exponent {maxE:.1f}")
        print(f"{basis}: worst rel {worst:.2e}")
