#!/usr/bin/env python3
"""End-to-end tour of the drop-in on one MI355X (python examples/demo_gpu.py).

Mirrors the model of the reference's README (README.md:38-71: SingleBendingPowerLaw -> approx -> ScalableGP -> logpdf)
on a synthetic irregular series, then the vectorised forms a sampler would call:
  1. one evaluation through the reference-shaped API (scalar `logl` drop-in),
  2. a batch of live points, theta -> log L in one call (approx on the device),
  3. value + gradient with respect to the sampled parameters (reverse mode on the device, chain rule through approx),
  4. posterior mean / standard deviation at new times and a prior draw for one posterior sample.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pioran_jl_amd as pj

rng = np.random.default_rng(1)
N = 2000
t = np.cumsum(0.05 + rng.exponential(0.95, N))
yerr = rng.uniform(0.007, 0.05, N)
f_min, f_max = 1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2
truth = dict(alpha1=0.82, f1=0.01, alpha2=3.3, variance=1.0, nu=1.0, mu=0.3)
R_true = pj.approx(pj.SingleBendingPowerLaw(truth["alpha1"], truth["f1"], truth["alpha2"]), f_min, f_max, 20, truth["variance"])
y = pj.rand(rng, pj.ScalableGP(truth["mu"], R_true)(t, yerr ** 2))          # a GP draw (simulated on the GPU)

# 1. the reference's call, one evaluation
f = pj.ScalableGP(truth["mu"], R_true)
print("log L at the truth:", pj.logpdf(f(t, truth["nu"] * yerr ** 2), y))

# 2. a batch of live points: theta -> log L
B = 4096
theta = np.column_stack([rng.uniform(0, 1.5, B), 10 ** rng.uniform(-3, 0, B), rng.uniform(1.5, 4, B)])
var, nu, mu = rng.lognormal(0, 0.5, B), rng.gamma(2, 0.5, B), rng.normal(0.3, 0.2, B)
ds = pj.Dataset(t, y, yerr ** 2)
ds.logpdf_theta(pj.SingleBendingPowerLaw, theta, var, f_min, f_max, 20, mu=mu, nu=nu)            # warm-up (table, buffers)
t0 = time.perf_counter()
ll, st = ds.logpdf_theta(pj.SingleBendingPowerLaw, theta, var, f_min, f_max, 20, mu=mu, nu=nu, return_status=True)
dt = time.perf_counter() - t0
print(f"{B} live points in {dt * 1e3:.1f} ms ({B / dt:.0f} evals/s), best log L = {np.nanmax(ll):.3f}, ok = {(st == 0).mean():.3f}")

# 3. value + gradient for a few chains
k = np.argsort(ll)[-4:]
g = ds.logpdf_theta_grad(pj.SingleBendingPowerLaw, theta[k], var[k], f_min, f_max, 20, mu=mu[k], nu=nu[k])
print("gradient wrt (alpha1, f1, alpha2 | variance | nu | mu) of the best point:",
      np.round(g["grad_theta"][-1], 3), round(g["grad_norm"][-1], 3), round(g["grad_nu"][-1], 3), round(g["grad_mu"][-1], 3))

# 4. posterior predictive pieces for the best point
b = k[-1]
Rb = pj.approx(pj.SingleBendingPowerLaw(*theta[b]), f_min, f_max, 20, var[b])
fp = pj.posterior(pj.ScalableGP(mu[b], Rb)(t, nu[b] * yerr ** 2), y)
tau = np.linspace(t[0], t[-1] + 50, 400)
m, s = pj.mean(fp, tau), pj.std(fp, tau)
print(f"posterior mean / std at {len(tau)} new times: mean in [{m.min():.2f}, {m.max():.2f}], std in [{s.min():.3f}, {s.max():.3f}]")
