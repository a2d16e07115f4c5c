#!/usr/bin/env python3
"""CPU (numpy): why the scan over the segments' elements is unstable, and which formulation of the combination is not.  The case: bench.py's long-series draw (theta[0] of
seed 4321, drawn for the N = 1e4 series) of SHO-20 on N = 65536 stamps — through the GPU scan alone 5e-4 off at 256 segments.  Elements per segment as the GPU forms them
(tools/time_parallel_proto.py), then a Kogge-Stone scan in several variants of  a_i (x) a_j , each compared boundary by boundary with the sequential walk.
usage: python tools/tp_scan_stability_proto.py [N = 16384] [segments = 64]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import bench
from oracle import oracle as O
from time_parallel_proto import StateSpace

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 64
draw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0_, y0_, e0_ = bench.synth_series(10000)
theta, _, _ = bench.synth_theta(4096, t0_, y0_, seed=4321)       # (the stream depends on the count: bench.py draws 4096)
t, y, e = bench.synth_series(N)
A_, B_, C_, D_, mu, nu = O.theta_to_coefs(theta[draw:draw + 1], t, 20, "SHO")
ss = StateSpace(A_[0], B_[0], C_, D_, t, y - mu[0], nu[0] * e ** 2)
R = ss.R
edges = np.linspace(0, N, nseg + 1).astype(int)
t0 = time.time()
cache = Path(f"/tmp/tp_proto_els_{N}_{nseg}_{draw}.npz")
if cache.exists():
    z_ = np.load(cache); els = [tuple(z_[f"e{p}_{k}"] for k in range(5)) for p in range(nseg - 1)]
else:
    els = [ss.segment_element(edges[p], edges[p + 1]) for p in range(nseg - 1)]
    np.savez(cache, **{f"e{p}_{k}": els[p][k] for p in range(nseg - 1) for k in range(5)})
print(f"N = {N}, {nseg} segments, R = {R}: elements in {time.time() - t0:.1f} s; |J| of a segment up to {max(np.abs(el[4]).max() for el in els):.1e}, |C| up to {max(np.abs(el[2]).max() for el in els):.1e}")
walk = [(np.zeros(R), ss.Pinf.copy())]
for p in range(nseg - 1):
    walk.append(ss.apply(els[p], *walk[-1]))
I = np.eye(R)

def combine(ei, ej, variant):
    Ai, bi, Ci, ei_, Ji = ei
    Aj, bj, Cj, ej_, Jj = ej
    W = I + Ci @ Jj
    sol = np.linalg.solve(W, np.column_stack([bi + Ci @ ej_, Ci, Ai]))
    z, Z, MA = sol[:, 0], sol[:, 1:R + 1], sol[:, R + 1:]
    b = Aj @ z + bj
    C = Aj @ Z @ Aj.T + Cj
    A = Aj @ MA
    v = ej_ - Jj @ bi
    if variant == "gpu":                       # the shipped formulas: products with J_j
        Jn = Ai.T @ (Jj @ MA) + Ji
        eta = Ai.T @ (v - Jj @ (Z @ v)) + ei_
    elif variant in ("transposed solve", "eta solved, J product", "J solved, eta product", "transposed solve + refinement"):
        # X = (I + J_j C_i)^-1 J_j and (I + J_j C_i)^-1 v as SOLUTIONS
        Wt = I + Jj @ Ci
        rhs = np.column_stack([Jj, v])
        X = np.linalg.solve(Wt, rhs)
        if variant.endswith("refinement"):
            res = rhs - (X + Jj @ (Ci @ X))
            X = X + np.linalg.solve(Wt, res)
        Jn = Ai.T @ X[:, :R] @ Ai + Ji if variant != "eta solved, J product" else Ai.T @ (Jj @ MA) + Ji
        eta = Ai.T @ X[:, R] + ei_ if variant != "J solved, eta product" else Ai.T @ (v - Jj @ (Z @ v)) + ei_
    elif variant == "solves, b through eta":   # additionally z = M (b_i + C_i eta_j) rewritten: M C_i = C_i M' (symmetric C, J):  z = M b_i + C_i (M' eta_j)
        Wt = I + Jj @ Ci
        X = np.linalg.solve(Wt, np.column_stack([Jj, v, ej_]))
        Jn = Ai.T @ X[:, :R] @ Ai + Ji
        eta = Ai.T @ X[:, R] + ei_
        z = np.linalg.solve(W, bi) + Ci @ X[:, R + 1]
        b = Aj @ z + bj
    elif variant == "factor of J":             # J_j = G G', (I + J_j C_i)^-1 J_j = G (I + G' C_i G)^-1 G' (an SPD inner matrix, eigenvalues >= 1), eta_j = G w
        lam, V = np.linalg.eigh(0.5 * (Jj + Jj.T))
        keep = lam > lam.max() * 1e-18
        G = V[:, keep] * np.sqrt(lam[keep])
        S = np.eye(G.shape[1]) + G.T @ Ci @ G
        w = np.linalg.lstsq(G, v, rcond=None)[0]
        Jn = Ai.T @ (G @ np.linalg.solve(S, G.T)) @ Ai + Ji
        eta = Ai.T @ (G @ np.linalg.solve(S, w)) + ei_
        # and M through Woodbury on the same inner matrix: M = I - C_i G S^-1 G'
        M = I - Ci @ G @ np.linalg.solve(S, G.T)
        z = M @ (bi + Ci @ ej_); Z = M @ Ci; MA = M @ Ai
        b = Aj @ z + bj; C = Aj @ Z @ Aj.T + Cj; A = Aj @ MA
    return A, b, 0.5 * (C + C.T), eta, 0.5 * (Jn + Jn.T)

prior = (np.zeros((R, R)), np.zeros(R), ss.Pinf.copy(), np.zeros(R), np.zeros((R, R)))
for variant in ("gpu", "transposed solve", "eta solved, J product", "J solved, eta product", "transposed solve + refinement", "solves, b through eta"):
    cur = [prior] + list(els)                  # scan index 0 = the prior, p = element of segment p - 1
    stride = 1
    while stride < nseg:
        nxt = list(cur)
        for p in range(stride, nseg):
            nxt[p] = combine(cur[p - stride], cur[p], variant)
        cur = nxt; stride *= 2
    worst = 0.0; worst_m = 0.0
    ld = q = 0.0
    for p in range(nseg):
        m_, P_ = (cur[p][1], cur[p][2])
        mw, Pw = walk[p]
        Sn = ss.s2[edges[p] - 1 if p else 0] + Pw[0::2, 0::2].sum()
        worst = max(worst, np.abs(P_ - Pw).max() / Sn); worst_m = max(worst_m, np.abs(m_ - mw).max() / np.sqrt(Sn))
        _, _, l_, q_, _ = ss.filter_range(m_, P_, edges[p], edges[p + 1])
        ld += l_; q += q_
    ll = -0.5 * ld - 0.5 * N * 1.8378770664093454836 - 0.5 * q
    if variant == "gpu":
        ldw = qw = 0.0
        for p in range(nseg):
            _, _, l_, q_, _ = ss.filter_range(walk[p][0], walk[p][1], edges[p], edges[p + 1])
            ldw += l_; qw += q_
        llw = -0.5 * ldw - 0.5 * N * 1.8378770664093454836 - 0.5 * qw
        print(f"   sequential walk: log L = {llw!r}")
    print(f"   scan, {variant:30s}: log L = {ll!r}  rel vs walk {abs(ll - llw) / abs(llw):.1e};  worst boundary state vs walk: |dP| / S {worst:.1e}, |dm| / sqrt(S) {worst_m:.1e}", flush=True)
