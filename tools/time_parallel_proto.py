#!/usr/bin/env python3
"""Prototype (numpy, CPU): a TIME-PARALLEL evaluation of the celerite log-likelihood — the reference's own benchmark is one
evaluation at a time (benchmark/benchmarks.jl:76-91), which the GPU runs as a serial chain of N steps (0.18 us per step at 40 rows: DESIGN.md section 10).

Form.  A celerite kernel is the covariance of a linear-Gaussian state-space model (two state components per term: decay x rotation), and the
recurrence of src/celerite_solver.jl:12-100 is its Kalman filter in other coordinates (S_n = P_inf - P_n, D_n = the innovation variance).  A Kalman filter
parallelises over time with the associative elements of Sarkka & Garcia-Fernandez (IEEE TAC 66, 2021: "Temporal parallelization of Bayesian
smoothers"): a = (A, b, C, eta, J) with  p(x_k | y_k, x_k-1) = N(A x_k-1 + b, C),  p(y_k | x_k-1) ~ N_info(eta, J), and
    a_i (x) a_j:  A = A_j M A_i,  b = A_j M (b_i + C_i eta_j) + b_j,  C = A_j M C_i A_j' + C_j,  M = (I + C_i J_j)^-1,
                  eta = A_i' M' (eta_j - J_j b_i) + eta_i,  J = A_i' M' J_j A_i + J_i.
Three phases:
  1  (parallel over P segments)  the element of each segment, composed step by step: a single step's J_j is rank one, so M is a Sherman-Morrison
     correction and every composition costs O(R^2) — about 30 R^2 flop per step against the recurrence's 5.5 R^2;
  2  (sequential over segments, or a log-depth scan)  the filtered state (m, P) at every segment boundary: one R x R solve and two products each;
  3  (parallel over segments)  the ordinary recurrence from each boundary state: log-determinant and quadratic form of the segment.
No inverse of the transition matrix appears anywhere (the linear-fractional form of the Riccati map needs e^{+c dt} and loses everything at the
high-frequency terms of the approx basis).

What this script reports: agreement of the three-phase evaluation with the C oracle on values the REFERENCE computed (tests/golden/ultranest_points.npz),
on synthetic bench-model draws at N = 1e4, and on the ill-conditioned draws of tests/golden/quad_truth.npz against their __float128 truth — and the
cost model for a GPU build.  usage: python tools/time_parallel_proto.py [segments=16]      (CPU only, ~10 minutes)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

LOG2PI = 1.8378770664093454836


class StateSpace:
    """Two state components per term j: F(dt) = e^{-c dt} Rot(d dt), h = (1, 0), P_inf = [[a, -b], [-b, a]]:
    h F(tau) P_inf h' = e^{-c tau} (a cos d tau + b sin d tau) = the term of src/acvf.jl's Celerite.  (The entry P_inf[1][1] is free — the
    likelihood only sees h F^k P_inf h' — and a keeps Q = P_inf - F P_inf F' positive semi-definite for the SHO / DRW bases.)"""

    def __init__(self, a, b, c, d, t, y, s2):
        self.a, self.b, self.c, self.d = (np.asarray(v, float) for v in (a, b, c, d))
        self.t, self.y, self.s2 = (np.asarray(v, float) for v in (t, y, s2))
        self.J = len(self.a)
        self.R = 2 * self.J
        self.h = np.zeros(self.R); self.h[0::2] = 1.0
        self.Pinf = np.zeros((self.R, self.R))
        for j in range(self.J):
            self.Pinf[2 * j, 2 * j] = self.Pinf[2 * j + 1, 2 * j + 1] = self.a[j]
            self.Pinf[2 * j, 2 * j + 1] = self.Pinf[2 * j + 1, 2 * j] = -self.b[j]
        dt = np.diff(self.t, prepend=self.t[0])
        self.ec = np.exp(-np.outer(dt, self.c))             # [N][J]
        self.co = self.ec * np.cos(np.outer(dt, self.d))
        self.si = self.ec * np.sin(np.outer(dt, self.d))

    def Fmul(self, n, X):
        """F_n X (rows in pairs), X: [R] or [R][k]"""
        co, si = self.co[n], self.si[n]
        if X.ndim == 2:
            co, si = co[:, None], si[:, None]
        out = np.empty_like(X)
        out[0::2] = co * X[0::2] - si * X[1::2]
        out[1::2] = si * X[0::2] + co * X[1::2]
        return out

    def FTmul(self, n, X):
        co, si = self.co[n], self.si[n]
        if X.ndim == 2:
            co, si = co[:, None], si[:, None]
        out = np.empty_like(X)
        out[0::2] = co * X[0::2] + si * X[1::2]
        out[1::2] = -si * X[0::2] + co * X[1::2]
        return out

    def Q(self, n):
        """P_inf - F_n P_inf F_n' (block diagonal: (1 - e^{-2 c dt}) P_inf_j, the rotation commutes with [[a, -b], [-b, a]]... only for b = 0;
        formed in general)"""
        FP = self.Fmul(n, self.Pinf)
        return self.Pinf - self.Fmul(n, FP.T).T

    # -- the ordinary filter from a boundary state: steps n0 .. n1 - 1 (phase 3; with the prior and the whole range: the sequential evaluation) --
    def filter_range(self, m, P, n0, n1):
        h = self.h
        logdet = quad = 0.0
        dmin = np.inf
        for n in range(n0, n1):
            if n > 0:
                m = self.Fmul(n, m)
                FP = self.Fmul(n, P)
                P = self.Fmul(n, FP.T).T + self.Q(n)
            Ph = P[:, 0::2].sum(axis=1)
            S = Ph[0::2].sum() + self.s2[n]
            v = self.y[n] - m[0::2].sum()
            logdet += np.log(abs(S)) if n > 0 else np.log(S)      # (src/celerite_solver.jl:126, 140: log D_1, log |D_n|)
            quad += v * v / S
            dmin = min(dmin, S)
            K = Ph / S
            m = m + K * v
            P = P - np.outer(K, Ph)
        return m, P, logdet, quad, dmin

    # -- phase 1: the element of steps n0 .. n1 - 1, composed step by step with rank-one corrections ---------------------------------------
    def segment_element(self, n0, n1):
        R, h = self.R, self.h
        A = np.eye(R); b = np.zeros(R); C = np.zeros((R, R)); eta = np.zeros(R); Jm = np.zeros((R, R))
        for n in range(n0, n1):
            if n == 0:     # the prior as "filtered state before step 0": F = I, Q = 0
                Qn = np.zeros((R, R)); g = h.copy()
                Fm = lambda X: X
            else:
                Qn = self.Q(n); g = self.FTmul(n, h)
                Fm = lambda X, n=n: self.Fmul(n, X)
            Qh = Qn[:, 0::2].sum(axis=1)
            s = Qh[0::2].sum() + self.s2[n]
            K = Qh / s
            yj = self.y[n]

            def Aj(X):          # (I - K h) F X
                FX = Fm(X)
                hFX = FX[0::2].sum(axis=0)
                return FX - (np.outer(K, hFX) if X.ndim == 2 else K * hFX)
            u = C @ g
            delta = s + g @ u
            gA = g @ A
            gb = g @ b
            A_new = Aj(A - np.outer(u, gA) / delta)
            b_new = Aj(b + u * (yj / s) - u * ((g @ (b + u * (yj / s))) / delta)) + K * yj
            Cm = C - np.outer(u, u) / delta
            C_new = Aj(Aj(Cm).T).T + (Qn - np.outer(K, Qh))
            Ag = A.T @ g
            eta = eta + Ag * ((yj - gb) / delta)
            Jm = Jm + np.outer(Ag, Ag) / delta
            A, b, C = A_new, b_new, 0.5 * (C_new + C_new.T)
        return A, b, C, eta, Jm

    # -- phase 2: a segment's element applied to the filtered state at its start ---------------------------------------------------------
    @staticmethod
    def apply(el, m, P):
        A, b, C, eta, Jm = el
        R = len(m)
        M = np.linalg.solve(np.eye(R) + P @ Jm, np.column_stack([m + P @ eta, P]))
        m2 = A @ M[:, 0] + b
        P2 = A @ M[:, 1:] @ A.T + C
        return m2, 0.5 * (P2 + P2.T)

    def logl_sequential(self):
        _, _, ld, q, dmin = self.filter_range(np.zeros(self.R), self.Pinf.copy(), 0, len(self.t))
        return -0.5 * ld - 0.5 * len(self.t) * LOG2PI - 0.5 * q

    def logl_time_parallel(self, nseg):
        N = len(self.t)
        edges = np.linspace(0, N, nseg + 1).astype(int)
        els = [self.segment_element(edges[p], edges[p + 1]) for p in range(nseg - 1)]         # phase 1 (independent)
        states = [(np.zeros(self.R), self.Pinf.copy())]
        for p in range(nseg - 1):                                                             # phase 2 (sequential, R^3 each)
            states.append(self.apply(els[p], *states[-1]))
        ld = q = 0.0
        for p in range(nseg):                                                                 # phase 3 (independent)
            _, _, l_, q_, _ = self.filter_range(states[p][0], states[p][1], edges[p], edges[p + 1])
            ld += l_; q += q_
        return -0.5 * ld - 0.5 * N * LOG2PI - 0.5 * q, states


def lane_form_tables(a, b, c, d, t, s2):
    """What the GPU kernels read per step (celerite_tp.hip): co, si = e^{-c dt} (cos, sin)(d dt), gam = -expm1(-2 c dt) per term; step 0 is the prior
    as 'filtered state before the first step' (F = I, Q = 0).  Q_j = [[a gam - 2 b co si, -b (gam + 2 si^2)], [., a gam + 2 b co si]] — no
    subtraction of nearly equal numbers at small c dt."""
    dt = np.diff(t, prepend=t[0])
    ec = np.exp(-np.outer(dt, c))
    co, si = ec * np.cos(np.outer(dt, d)), ec * np.sin(np.outer(dt, d))
    gam = -np.expm1(-2.0 * np.outer(dt, c))
    return co, si, gam


def lane_form_element(a, b, c, d, t, y, s2, n0, n1):
    """The segment element in the form the GPU kernel composes it (A kept transposed; every product a rotation of adjacent rows / columns, a
    rank-one update, or a matrix-vector sum): checked here against StateSpace.segment_element."""
    J = len(a); R = 2 * J
    co, si, gam = lane_form_tables(a, b, c, d, t, s2)
    par = np.arange(R) & 1; term = np.arange(R) >> 1
    At = np.eye(R); C = np.zeros((R, R)); Jm = np.zeros((R, R)); bv = np.zeros(R); eta = np.zeros(R)
    ev = par == 0

    def rot_cols(X, al, be):      # X F': column r <- al_r X[:, r] + be_r X[:, r ^ 1]
        return X * al[None, :] + X[:, np.arange(R) ^ 1] * be[None, :]

    def rot_rows(X, al, be):
        return X * al[:, None] + X[np.arange(R) ^ 1, :] * be[:, None]
    for n in range(n0, n1):
        al = co[n][term]; be = np.where(ev, -si[n][term], si[n][term])
        g = np.where(ev, co[n][term], -si[n][term])
        q00 = a * gam[n] - 2 * b * co[n] * si[n]; q01 = -b * (gam[n] + 2 * si[n] ** 2); q11 = a * gam[n] + 2 * b * co[n] * si[n]
        Qh = np.where(ev, q00[term], q01[term])
        s = q00.sum() + s2[n]
        K = Qh / s
        yj = y[n]
        u = C @ g; delta = s + g @ u; ag = At @ g; gb = g @ bv
        Xt = At - np.outer(ag, u) / delta
        XtF = rot_cols(Xt, al, be)
        hFX = XtF[:, ev].sum(axis=1)
        At_new = XtF - np.outer(hFX, K)
        bb = bv + u * (yj / s); bb = bb - u * ((g @ bb) / delta)
        Fb = al * bb + be * bb[np.arange(R) ^ 1]
        b_new = Fb - K * Fb[ev].sum() + K * yj
        Cm = C - np.outer(u, u) / delta
        Y = rot_cols(rot_rows(Cm, al, be), al, be)
        Yh = Y[:, ev].sum(axis=1); hYh = Yh[ev].sum()
        Qd = np.zeros((R, R))
        for j in range(J):
            Qd[2 * j, 2 * j] = q00[j]; Qd[2 * j + 1, 2 * j + 1] = q11[j]; Qd[2 * j, 2 * j + 1] = Qd[2 * j + 1, 2 * j] = q01[j]
        C = Y - np.outer(K, Yh) - np.outer(Yh, K) + np.outer(K, K) * hYh + Qd - np.outer(K, Qh)
        eta = eta + ag * ((yj - gb) / delta)
        Jm = Jm + np.outer(ag, ag) / delta
        At, bv = At_new, b_new
    return At.T, bv, C, eta, Jm


def rel(a, b):
    return abs(a - b) / abs(b)


def main():
    nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(f"# time-parallel evaluation, {nseg} segments, against the C oracle (fp64) and, where there is one, the __float128 truth")
    # (1) values the reference computed: stored ultranest run, N = 242, SHO-20 (tests/test_oracle.py::_un_inputs)
    un = np.load(ROOT / "tests" / "golden" / "ultranest_points.npz")
    sys.path.insert(0, str(ROOT / "tests"))
    from test_oracle import _un_inputs  # noqa: E402
    idx = np.linspace(0, len(un["logl"]) - 1, 150).astype(int)
    e_seq, e_par, e_ref = [], [], []
    t0 = time.time()
    for i in idx:
        a, b, c, d, t, y, s2 = _un_inputs(un, i)
        ss = StateSpace(a, b, c, d, t, y, s2)
        o = O.logl(a, b, c, d, t, y, s2)
        ls = ss.logl_sequential(); lp, _ = ss.logl_time_parallel(min(nseg, 8))
        e_seq.append(rel(ls, o)); e_par.append(rel(lp, o)); e_ref.append(rel(lp, un["logl"][i]))
    print(f"reference-computed values (ultranest run, N = 242, 20 terms, {len(idx)} of {len(un['logl'])} points, 8 segments; {time.time() - t0:.0f} s):")
    print(f"  state-space filter, sequential, vs oracle: max {max(e_seq):.2e} median {np.median(e_seq):.2e}")
    print(f"  time-parallel vs oracle: max {max(e_par):.2e} median {np.median(e_par):.2e};  vs the reference's stored value: max {max(e_ref):.2e}")
    # (2) the bench model at full length
    t, y, yerr = O.synthetic_series(10000, seed=1234)
    th = O.synthetic_theta(4, t, y, seed=4321)
    for basis in ("SHO", "DRWCelerite"):
        A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20 if basis == "SHO" else 10, basis)
        for i in range(2):
            t0 = time.time()
            ss = StateSpace(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2)
            o = O.logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2)
            lp, _ = ss.logl_time_parallel(nseg)
            print(f"bench series N = 1e4, {basis}-{A.shape[1]} draw {i}: time-parallel vs oracle {rel(lp, o):.2e}   (log L = {o:.6f}; {time.time() - t0:.0f} s)", flush=True)
    # (3) ill-conditioned draws with their __float128 truth (N = 150 and 1000: every bin; N = 1e4: a few)
    q = np.load(ROOT / "tests" / "golden" / "quad_truth.npz")
    for N, take in ((150, 120), (1000, 40)):
        tag = f"n{N}"
        t, y, yerr = q[f"{tag}_t"], q[f"{tag}_y"], q[f"{tag}_yerr"]
        A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
        truth, ratio, orc = q[f"{tag}_truth"], q[f"{tag}_ratio"], q[f"{tag}_oracle_fp64"]
        order = np.argsort(ratio)
        pick = order[np.linspace(0, len(order) - 1, take).astype(int)]
        ep, eo = [], []
        for i in pick:
            ss = StateSpace(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * yerr ** 2)
            lp, _ = ss.logl_time_parallel(min(nseg, N // 16))
            ep.append(rel(lp, truth[i])); eo.append(rel(orc[i], truth[i]))
        ep, eo, r = np.array(ep), np.array(eo), ratio[pick]
        print(f"ill-conditioned draws, N = {N} ({take} of {len(truth)}), relative deviation from the __float128 truth, max / median per bin of ratio = nu min(sigma2) / sum(a):")
        edges = [0, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5]
        for lo, hi in zip(edges, edges[1:]):
            msk = (r >= lo) & (r < hi)
            if msk.any():
                print(f"  [{lo:.0e}, {hi:.0e}): {msk.sum():3d} draws | time-parallel {np.nanmax(ep[msk]):.1e} / {np.nanmedian(ep[msk]):.1e} | fp64 oracle {eo[msk].max():.1e} / {np.median(eo[msk]):.1e}", flush=True)


if __name__ == "__main__":
    main()
