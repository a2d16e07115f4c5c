#!/usr/bin/env python3
"""Hamiltonian Monte Carlo on the GPU likelihood + gradient (python examples/hmc_demo_gpu.py).

What Turing/NUTS does with the reference through ForwardDiff Duals (docs/src/turing.md; test/test_likelihood.jl:55-60),
here with the device gradient: several chains advance in lock-step, ONE batched value-and-gradient call per leapfrog step.
Model: the README's (SingleBendingPowerLaw -> approx -> ScalableGP), sampled in unconstrained coordinates
q = (alpha1, log f1, alpha2, log variance, log nu, mu) with flat priors on a box (a demo, not a recommended prior).
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pioran_jl_amd as pj

rng = np.random.default_rng(3)
N = 1500
t = np.cumsum(0.05 + rng.exponential(0.95, N))
yerr = rng.uniform(0.007, 0.05, N)
f_min, f_max = 1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2
truth = np.array([0.6, np.log(0.02), 3.0, np.log(1.0), np.log(1.0), 0.3])
R_true = pj.approx(pj.SingleBendingPowerLaw(truth[0], np.exp(truth[1]), truth[2]), f_min, f_max, 20, np.exp(truth[3]))
y = pj.rand(rng, pj.ScalableGP(truth[5], R_true)(t, yerr ** 2))
ds = pj.Dataset(t, y, yerr ** 2)
lo = np.array([0.0, np.log(f_min), 1.5, np.log(0.05), np.log(0.2), -2.0])
hi = np.array([1.5, np.log(f_max), 4.0, np.log(20.0), np.log(5.0), 2.0])

def logp_and_grad(q):
    """q (C, 6) -> log posterior (C,), gradient (C, 6); -inf outside the box."""
    inside = np.all((q > lo) & (q < hi), axis=1)
    qq = np.where(inside[:, None], q, truth)
    theta = np.column_stack([qq[:, 0], np.exp(qq[:, 1]), qq[:, 2]])
    var, nu, mu = np.exp(qq[:, 3]), np.exp(qq[:, 4]), qq[:, 5]
    g = ds.logpdf_theta_grad(pj.SingleBendingPowerLaw, theta, var, f_min, f_max, 20, mu=mu, nu=nu)
    grad = np.column_stack([g["grad_theta"][:, 0], g["grad_theta"][:, 1] * theta[:, 1], g["grad_theta"][:, 2],
                            g["grad_norm"] * var, g["grad_nu"] * nu, g["grad_mu"]])
    lp = np.where(inside & (g["status"] == 0), g["logl"], -np.inf)
    return lp, np.where(np.isfinite(lp)[:, None], grad, 0.0)

C = 8                                             # chains
q = truth + 0.05 * rng.standard_normal((C, 6))
lp, gr = logp_and_grad(q)
# diagonal mass matrix from the curvature of log p along each coordinate at the start (conditional widths)
h = 1e-3
scale = np.empty(6)
for k in range(6):
    e = np.zeros(6); e[k] = h
    curv = (logp_and_grad(q + e)[0] - 2 * lp + logp_and_grad(q - e)[0]) / h ** 2
    scale[k] = 1 / np.sqrt(np.median(np.abs(curv)))
eps, L, n_iter = 0.35, 12, 150
acc = 0; samples = []; calls = 0
t0 = time.perf_counter()
for it in range(n_iter):
    p0 = rng.standard_normal((C, 6))
    qn, pn, lpn, grn = q.copy(), p0.copy(), lp.copy(), gr.copy()
    for _ in range(L):                            # leapfrog in the scaled coordinates
        pn = pn + 0.5 * eps * grn * scale
        qn = qn + eps * pn * scale
        lpn, grn = logp_and_grad(qn); calls += 1
        pn = pn + 0.5 * eps * grn * scale
    dH = (lpn - 0.5 * (pn ** 2).sum(1)) - (lp - 0.5 * (p0 ** 2).sum(1))
    ok = np.log(rng.random(C)) < dH
    q[ok], lp[ok], gr[ok] = qn[ok], lpn[ok], grn[ok]
    acc += ok.sum()
    if it >= n_iter // 3: samples.append(q.copy())
dt = time.perf_counter() - t0
S = np.concatenate(samples)
names = ["alpha1", "log f1", "alpha2", "log var", "log nu", "mu"]
print(f"{C} chains x {n_iter} HMC iterations, {calls} batched value+gradient calls in {dt:.1f} s "
      f"({dt / calls * 1e3:.1f} ms per call of {C} chains), acceptance {acc / (C * n_iter):.2f}")
for k, nm in enumerate(names):
    print(f"  {nm:8s} truth {truth[k]:7.3f}   posterior {S[:, k].mean():7.3f} +- {S[:, k].std():.3f}")
