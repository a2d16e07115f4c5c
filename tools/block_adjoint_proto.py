"""Numerical prototype (numpy, never shipped) of the REVERSE pass of the windowed form (tools/block_proto.py is the forward): the
gradient of log L with respect to (a_j, b_j, c_j, d_j, y_n, sigma2_n), one window of K steps at a time, written with the window's
inverse K = Sigma^-1 instead of the adjoint of the LDL' factorisation:
  forward    M = U T;  S = A - M U';  X = V^ - M o cK;  Q = K X (K = S^-1);  T' = (cK cK') o T + X' Q;
             l += -1/2 log det S - 1/2 x_y' K x_y
  reverse    X- = 2 Q T-' (- Q[:, y] in the y column);  S- = -1/2 K - Q T-' Q' + 1/2 q_y q_y';
             T- = (cK cK') o T-' + sym(U' M-);  M- = -X- o cK - S- U;  U- = -S- M + M- T
  and the parameter adjoints from (U-, V^- = X-, A- = S-, cK-).
Checked against the complex-step oracle (oracle.logl_grad).  Usage: python tools/block_adjoint_proto.py"""
import sys
from pathlib import Path
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import oracle as O  # noqa: E402


def rows_of(a, b, c, d):
    J = len(a); R = 2 * J + 1
    al = np.zeros(R); be = np.zeros(R); cc = np.zeros(R); dd = np.zeros(R); sinrow = np.zeros(R, bool); term = np.full(R, -1)
    al[0:2 * J:2] = a; be[0:2 * J:2] = b; al[1:2 * J:2] = a; be[1:2 * J:2] = -b
    cc[0:2 * J:2] = c; cc[1:2 * J:2] = c; dd[0:2 * J:2] = d; dd[1:2 * J:2] = d; sinrow[1:2 * J:2] = True
    term[0:2 * J:2] = np.arange(J); term[1:2 * J:2] = np.arange(J)
    return al, be, cc, dd, sinrow, term


def window_inputs(m, n1, a, b, c, d, t, y, s2, rows):
    al, be, cc, dd, sinrow, term = rows
    R = len(al); J = len(a); k = n1 - m
    tt = t[m:n1]
    co = np.cos(np.outer(tt, dd)); si = np.sin(np.outer(tt, dd))
    v = np.where(sinrow, si, co); x = np.where(sinrow, co, si)
    v[:, R - 1] = y[m:n1]; x[:, R - 1] = 0.0
    tb = t[m - 1] if m > 0 else t[0]; te = tt[-1]
    C = np.exp(-np.outer(tt - tb, cc)); Cend = np.exp(-np.outer(te - tt, cc)); cK = np.exp(-cc * (te - tb))
    cj = np.cos(np.outer(tt, d)); sj = np.sin(np.outer(tt, d))
    tau = tt[:, None] - tt[None, :]                                       # tau[n][j] = t_n - t_j
    later = tau >= 0
    E = []                                                                # per term: (dec, cosd, sind) as k x k (tau >= 0 part used)
    A = np.zeros((k, k))
    for j in range(J):
        cd = np.outer(cj[:, j], cj[:, j]) + np.outer(sj[:, j], sj[:, j])
        sd = np.outer(sj[:, j], cj[:, j]) - np.outer(cj[:, j], sj[:, j])  # sin(d (t_n - t_j)) at [n][j]
        dec = np.exp(-c[j] * np.abs(tau))
        sds = np.where(later, sd, -sd)
        E.append((dec, cd, sds))
        A += dec * (a[j] * cd + b[j] * sds)
    A[np.diag_indices(k)] = a.sum() + s2[m:n1]
    return dict(k=k, tt=tt, tb=tb, te=te, v=v, x=x, C=C, Cend=Cend, cK=cK, A=A, E=E, abstau=np.abs(tau))


def forward(a, b, c, d, t, y, s2, K=16):
    rows = rows_of(a, b, c, d)
    al, be = rows[0], rows[1]
    R = len(al); N = len(t)
    T = np.zeros((R, R)); ll = 0.0; store = []
    for m in range(0, N, K):
        n1 = min(N, m + K)
        w = window_inputs(m, n1, a, b, c, d, t, y, s2, rows)
        U = (al * w["v"] + be * w["x"]) * w["C"]
        M = U @ T
        S = w["A"] - M @ U.T
        Vh = w["v"] * w["Cend"]
        X = Vh - M * w["cK"]
        Kinv = np.linalg.inv(S)
        Q = Kinv @ X
        sign, logdet = np.linalg.slogdet(S)
        ll += -0.5 * logdet - 0.5 * X[:, R - 1] @ Q[:, R - 1]
        store.append(dict(m=m, n1=n1, T=T.copy(), U=U, M=M, X=X, Q=Q, Kinv=Kinv, Vh=Vh))
        T = T * np.outer(w["cK"], w["cK"]) + X.T @ Q
    return ll - 0.5 * N * np.log(2 * np.pi), store, rows


def reverse(a, b, c, d, t, y, s2, store, rows, K=16):
    al, be, cc, dd, sinrow, term = rows
    R = len(al); J = len(a); N = len(t)
    Tb = np.zeros((R, R))
    g_al = np.zeros(R); g_be = np.zeros(R); g_crow = np.zeros(R); g_drow = np.zeros(R)
    g_a = np.zeros(J); g_b = np.zeros(J); g_c = np.zeros(J); g_d = np.zeros(J); g_suma = 0.0
    g_y = np.zeros(N); g_s2 = np.zeros(N)
    for st in reversed(store):
        m, n1 = st["m"], st["n1"]
        w = window_inputs(m, n1, a, b, c, d, t, y, s2, rows)
        T, U, M, X, Q, Kinv, Vh = st["T"], st["U"], st["M"], st["X"], st["Q"], st["Kinv"], st["Vh"]
        cK = w["cK"]; tt, tb, te = w["tt"], w["tb"], w["te"]
        QT = Q @ Tb                                              # k x R
        Xb = 2.0 * QT
        Xb[:, R - 1] -= Q[:, R - 1]
        qy = Q[:, R - 1]
        Sb = -0.5 * Kinv - QT @ Q.T + 0.5 * np.outer(qy, qy)
        Sb = 0.5 * (Sb + Sb.T)
        cKb = 2.0 * np.einsum("ij,j,ij->i", Tb, cK, T) - np.einsum("ni,ni->i", Xb, M)
        Mb = -Xb * cK - Sb @ U
        Ub = -Sb @ M + Mb @ T
        UM = U.T @ Mb
        Tb = Tb * np.outer(cK, cK) + 0.5 * (UM + UM.T)
        # ---- parameters ----
        v, x, C, Cend = w["v"], w["x"], w["C"], w["Cend"]
        g_al += np.einsum("ni,ni,ni->i", Ub, C, v)
        g_be += np.einsum("ni,ni,ni->i", Ub, C, x)
        g_crow -= np.einsum("ni,ni,n->i", Ub, U, tt - tb)
        g_crow -= np.einsum("ni,ni,n->i", Xb, Vh, te - tt)
        g_crow -= cKb * cK * (te - tb)
        vb = Ub * C * al + Xb * Cend
        xb = Ub * C * be
        sgn = np.where(sinrow, 1.0, -1.0)
        g_drow += sgn * np.einsum("n,ni->i", tt, vb * x - xb * v)
        g_y[m:n1] += Xb[:, R - 1]
        g_s2[m:n1] += np.diag(Sb)
        g_suma += np.trace(Sb)
        off = Sb - np.diag(np.diag(Sb))
        for j in range(J):
            dec, cd, sds = w["E"][j]
            g_a[j] += np.sum(off * dec * cd)
            g_b[j] += np.sum(off * dec * sds)
            g_c[j] -= np.sum(off * dec * (a[j] * cd + b[j] * sds) * w["abstau"])
            # d/dd of (a cos(d tau) + b sin(d |tau|)) = |tau| (-a sin(d |tau|) + b cos(d tau)), with sin(d|tau|) = sds
            g_d[j] += np.sum(off * dec * w["abstau"] * (-a[j] * sds + b[j] * cd))
    for i in range(R - 1):
        j = term[i]
        g_a[j] += g_al[i]
        g_b[j] += -g_be[i] if sinrow[i] else g_be[i]
        g_c[j] += g_crow[i]
        g_d[j] += g_drow[i]
    g_a += g_suma
    return dict(grad_a=g_a, grad_b=g_b, grad_c=g_c, grad_d=g_d, grad_y=g_y, grad_sigma2=g_s2)


if __name__ == "__main__":
    rng = np.random.default_rng(7)
    worst = 0.0
    for N, J in ((50, 2), (100, 3), (37, 5)):
        t = np.cumsum(rng.uniform(0.05, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        a = rng.uniform(0.1, 2.0, J); b = rng.uniform(-0.05, 0.05, J) * a; c = rng.uniform(0.05, 2.0, J); d = rng.uniform(0.1, 3.0, J)
        ll, store, rows = forward(a, b, c, d, t, y, s2)
        ref = O.logl(a, b, c, d, t, y, s2)
        g = reverse(a, b, c, d, t, y, s2, store, rows)
        og = O.logl_grad(a, b, c, d, t, y, s2, series=True, cd=True)
        print(f"N={N} J={J}: logl {ll:.10f} oracle {ref:.10f}")
        for k in ("grad_a", "grad_b", "grad_c", "grad_d", "grad_y", "grad_sigma2"):
            scale = np.max(np.abs(og[k])) + 1e-300
            err = np.max(np.abs(g[k] - og[k])) / scale
            worst = max(worst, err)
            print(f"   {k:12s} max err / scale {err:.2e}")
    print("worst", worst)
