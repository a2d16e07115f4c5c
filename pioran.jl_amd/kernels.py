"""Kernel containers of the hot path — host-side mirror of the reference's types.

  Celerite / SHO / Exp            src/Celerite.jl:19-44, src/SHO.jl:19-56, src/Exp.jl:19-38
  SumOfCelerite / SumOfSemiSeparable, `+`, scaling, celerite_coefs     src/acvf.jl:16-156

Only what feeds the solvers is kept: each kernel knows its (a, b, c, d) and its closed-form k(tau).
"""
from __future__ import annotations

import math

import numpy as np


class SemiSeparable:
    """abstract type SemiSeparable (src/acvf.jl:8)."""

    def celerite_coefs(self):
        raise NotImplementedError

    def kappa(self, tau):
        raise NotImplementedError

    def __call__(self, x, y):
        # KernelFunctions.SimpleKernel with the Euclidean metric (src/acvf.jl:131,135)
        return self.kappa(np.abs(np.asarray(x, float) - np.asarray(y, float)))

    def __add__(self, other):
        return _add(self, other)

    def __rmul__(self, number):
        return self.scaled(float(number))

    def __mul__(self, number):
        return self.scaled(float(number))


class Celerite(SemiSeparable):
    """Celerite(a, b, c, d): k(tau) = exp(-c tau) (a cos(d tau) + b sin(d tau))   src/Celerite.jl:19-44."""

    def __init__(self, a, b, c, d):
        self.a, self.b, self.c, self.d = a, b, c, d

    def celerite_coefs(self):
        return [self.a, self.b, self.c, self.d]

    def kappa(self, tau):
        tau = np.asarray(tau, float)
        return np.exp(-self.c * tau) * (self.a * np.cos(self.d * tau) + self.b * np.sin(self.d * tau))

    def scaled(self, number):
        return Celerite(number * self.a, number * self.b, self.c, self.d)

    def __eq__(self, o):
        return isinstance(o, Celerite) and (self.a, self.b, self.c, self.d) == (o.a, o.b, o.c, o.d)


class SHO(SemiSeparable):
    """SHO(A, w0, Q)   src/SHO.jl:19-56."""

    def __init__(self, A, w0, Q):
        self.A, self.w0, self.Q = A, w0, Q

    def celerite_coefs(self):
        if self.Q == 1 / math.sqrt(2):
            c = math.sqrt(2) / 2 * self.w0
            return [self.A, self.A, c, c]
        raise ValueError("SHO with Q≠1/√2 not implemented yet")  # src/SHO.jl:39

    def kappa(self, tau):
        tau = np.asarray(tau, float)
        term1 = self.A * np.exp(-self.w0 * tau / self.Q / 2)
        eta = math.sqrt(abs(1 - 1 / (4 * self.Q ** 2)))
        if self.Q == 1 / 2:
            return term1 * 2 * (1 + self.w0 * tau)
        if self.Q >= 1 / 2:
            return term1 * (np.cos(eta * self.w0 * tau) + np.sin(eta * self.w0 * tau) / (2 * eta * self.Q))
        return term1 * (np.cosh(eta * self.w0 * tau) + np.sinh(eta * self.w0 * tau) / (2 * eta * self.Q))

    def scaled(self, number):
        return SHO(number * self.A, self.w0, self.Q)


class Exp(SemiSeparable):
    """Exp(A, alpha): k(tau) = A/2 exp(-alpha tau)   src/Exp.jl:19-38."""

    def __init__(self, A, alpha):
        self.A, self.alpha = A, alpha

    def celerite_coefs(self):
        return [self.A / 2, 0.0, self.alpha, 0.0]

    def kappa(self, tau):
        return self.A / 2 * np.exp(-self.alpha * np.asarray(tau, float))

    def scaled(self, number):
        return Exp(number * self.A, self.alpha)


class SumOfTerms(SemiSeparable):
    def kappa(self, tau):
        # src/acvf.jl:138-140
        return sum(k.kappa(tau) for k in self.cov)


class SumOfSemiSeparable(SumOfTerms):
    """src/acvf.jl:16-22."""

    def __init__(self, cov, a, b, c, d):
        self.cov = list(cov)
        self.a = np.asarray(a, float); self.b = np.asarray(b, float)
        self.c = np.asarray(c, float); self.d = np.asarray(d, float)

    def celerite_coefs(self):
        # src/acvf.jl:118-128: re-read from the member kernels
        co = np.array([k.celerite_coefs() for k in self.cov], float)
        return co[:, 0].copy(), co[:, 1].copy(), co[:, 2].copy(), co[:, 3].copy()

    def scaled(self, number):
        # src/acvf.jl:143-151
        return SumOfSemiSeparable([k.scaled(number) for k in self.cov], number * self.a, number * self.b, self.c,
                                  self.d)


class SumOfCelerite(SumOfTerms):
    """SumOfCelerite(a, b, c, d)   src/acvf.jl:35-53."""

    def __init__(self, a, b, c, d):
        self.a = np.asarray(a, float); self.b = np.asarray(b, float)
        self.c = np.asarray(c, float); self.d = np.asarray(d, float)
        if not (len(self.a) == len(self.b) == len(self.c) == len(self.d)):
            raise ValueError("a, b, c, d must have the same length")

    @property
    def cov(self):
        return [Celerite(*x) for x in zip(self.a, self.b, self.c, self.d)]

    def celerite_coefs(self):
        return self.a, self.b, self.c, self.d

    def kappa(self, tau):
        tau = np.asarray(tau, float)[..., None]
        return np.sum(np.exp(-self.c * tau) * (self.a * np.cos(self.d * tau) + self.b * np.sin(self.d * tau)), axis=-1)

    def scaled(self, number):
        return SumOfCelerite(number * self.a, number * self.b, self.c, self.d)


class CARMA(SemiSeparable):
    """CARMA(p, q, r_alpha, beta[, norm[, is_integrated_power]])   src/CARMA.jl:20-43.

    r_alpha: the p roots of the autoregressive polynomial, complex-conjugate pairs first (for odd p the last root is
    real); beta: the q + 1 moving-average coefficients.  On the likelihood path a CARMA kernel is only its celerite
    coefficients (`log_likelihood(::CARMA, ...)`, src/celerite_solver.jl:272-282)."""

    def __init__(self, p, q, r_alpha, beta, norm=1.0, is_integrated_power=True):
        if isinstance(norm, bool):   # CARMA(p, q, r_alpha, beta, is_integrated_power)   src/CARMA.jl:42
            norm, is_integrated_power = 1.0, norm
        p, q = int(p), int(q)
        r_alpha = np.atleast_1d(np.asarray(r_alpha, dtype=complex))
        beta = np.atleast_1d(np.asarray(beta, dtype=float))
        if p < 1 or q < 0:
            raise ValueError("The order of the autoregressive and moving average polynomials must be positive")
        if q > p:
            raise ValueError("The order of the moving average polynomial must be less than or equal to the order of the "
                             "autoregressive polynomial")
        if len(r_alpha) != p:
            raise ValueError("The length of the roots of the autoregressive polynomial must be equal to the order of the "
                             "autoregressive polynomial")
        if len(beta) != q + 1:
            raise ValueError("The length of the moving average coefficients must be equal to q + 1")
        self.p, self.q, self.r_alpha, self.beta = p, q, r_alpha, beta
        self.norm, self.is_integrated_power = float(norm), bool(is_integrated_power)

    def celerite_coefs(self):
        """CARMA_celerite_coefs (src/CARMA.jl:98-143): one celerite term per conjugate pair of roots (every second
        root), a real term for the unpaired last root of an odd p."""
        r_all, beta, p = self.r_alpha, self.beta, self.p
        rk = r_all[0::2]                                   # one root of each pair (+ the real one)
        J = len(rk)
        powers = np.arange(len(beta))
        num = (beta * rk[:, None] ** powers).sum(axis=1) * (beta * (-rk[:, None]) ** powers).sum(axis=1)
        frac = -num / rk.real
        for k in range(J):
            others = r_all[r_all != rk[k]]
            frac[k] /= np.prod((others - rk[k]) * (np.conj(others) + rk[k]))
        a, b, c, d = 2 * frac.real, 2 * frac.imag, -rk.real, -rk.imag
        if p % 2 == 1:                                     # the last root is real: a single exponential term
            a[-1], b[-1], d[-1] = frac[-1].real, 0.0, 0.0
        scale = self.norm / a.sum() if self.is_integrated_power else self.norm
        return a * scale, b * scale, c, d

    def celerite_repr(self):
        """celerite_repr(cov::CARMA)   src/CARMA.jl:57-72."""
        return SumOfCelerite(*self.celerite_coefs())

    def kappa(self, tau):
        a, b, c, d = self.celerite_coefs()
        tau = np.abs(np.asarray(tau, float))[..., None]
        return (np.exp(-c * tau) * (a * np.cos(d * tau) + b * np.sin(d * tau))).sum(-1)

    def scaled(self, number):
        return CARMA(self.p, self.q, self.r_alpha, self.beta, self.norm * number, self.is_integrated_power)


def celerite_coefs(cov: SemiSeparable):
    return cov.celerite_coefs()


def ScaledKernel(cov: SemiSeparable, number: float = 1.0):
    return cov.scaled(number)


def _add(cov1: SemiSeparable, cov2: SemiSeparable) -> SumOfSemiSeparable:
    """`+` of src/acvf.jl:60-111; the concatenation order is the reference's (pinned by test/test_acvf.jl:26-33)."""
    a1, b1, c1, d1 = (np.atleast_1d(np.asarray(x, float)) for x in cov1.celerite_coefs())
    a2, b2, c2, d2 = (np.atleast_1d(np.asarray(x, float)) for x in cov2.celerite_coefs())
    s1, s2 = isinstance(cov1, SumOfSemiSeparable), isinstance(cov2, SumOfSemiSeparable)
    if not s1 and not s2:
        cov = [cov1, cov2]
        cat = lambda x, y: np.concatenate([x, y])
    elif s1 and s2:
        cov = cov1.cov + cov2.cov
        cat = lambda x, y: np.concatenate([x, y])
    elif s1:
        cov = cov1.cov + [cov2]
        cat = lambda x, y: np.concatenate([x, y])
    else:  # src/acvf.jl:101-108: the sum's own terms first, then cov1's
        cov = [cov1] + cov2.cov
        cat = lambda x, y: np.concatenate([y, x])
    return SumOfSemiSeparable(cov, cat(a1, a2), cat(b1, b2), cat(c1, c2), cat(d1, d2))
