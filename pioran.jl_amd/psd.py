"""PSD -> celerite coefficients (host side of the path; src/psd.jl:15-27,73-112,214-289,301-324,375-395).

Stays on the host, as in the reference (BASELINE.json north_star): a J x J solve and 2J log/atan per
draw.  `approx` is the scalar reference signature; `approx_batch` is the vectorised front-end that
turns theta[B, .] into the (A, Bc)[B, J] + shared (c, d)[J] arrays the batched solver takes.

PSD model closed forms are Tonari.jl's (un-vendored dependency, Project.toml:46), pinned by the
reference's test/test_psd.jl:3-13.
"""
from __future__ import annotations

import math

import numpy as np

from .kernels import SumOfCelerite


class PowerSpectralDensity:
    def __add__(self, other):
        return SumOfPowerSpectralDensity(_flatten(self) + _flatten(other))


class ContinuumPowerSpectrum(PowerSpectralDensity):
    pass


class SingleBendingPowerLaw(ContinuumPowerSpectrum):
    """(f/f1)^-a1 / (1 + (f/f1)^(a2-a1))   test/test_psd.jl:3-7."""

    def __init__(self, alpha1, f1, alpha2):
        self.alpha1, self.f1, self.alpha2 = alpha1, f1, alpha2

    def __call__(self, f):
        f = np.asarray(f, float)
        return (f / self.f1) ** (-self.alpha1) / (1 + (f / self.f1) ** (self.alpha2 - self.alpha1))


class DoubleBendingPowerLaw(ContinuumPowerSpectrum):
    """test/test_psd.jl:9-13."""

    def __init__(self, alpha1, f1, alpha2, f2, alpha3):
        self.alpha1, self.f1, self.alpha2, self.f2, self.alpha3 = alpha1, f1, alpha2, f2, alpha3

    def __call__(self, f):
        f = np.asarray(f, float)
        return ((f / self.f1) ** (-self.alpha1) / (1 + (f / self.f1) ** (self.alpha2 - self.alpha1))
                / (1 + (f / self.f2) ** (self.alpha3 - self.alpha2)))


class QPO(PowerSpectralDensity):
    """QPO(S0, f0, Q): a PSD *feature*; enters the kernel as one extra celerite term (src/psd.jl:15-27)."""

    def __init__(self, S0, f0, Q):
        self.S0, self.f0, self.Q = S0, f0, Q


class SumOfPowerSpectralDensity(PowerSpectralDensity):
    def __init__(self, parts):
        self.parts = list(parts)


def _flatten(p):
    return list(p.parts) if isinstance(p, SumOfPowerSpectralDensity) else [p]


def separate_psd(psd_model):
    """Tonari.separate_psd as used at src/psd.jl:221: (continuum or None, features list or None)."""
    parts = _flatten(psd_model)
    cont = [p for p in parts if isinstance(p, ContinuumPowerSpectrum)]
    feat = [p for p in parts if not isinstance(p, ContinuumPowerSpectrum)]
    if len(cont) > 1:
        raise ValueError("only one continuum component is supported")
    return (cont[0] if cont else None), (feat if feat else None)


def convert_feature(psd_feature):
    """src/psd.jl:15-27."""
    if isinstance(psd_feature, QPO):
        delta = math.sqrt(4 * psd_feature.Q ** 2 - 1)
        w0 = 2 * math.pi * psd_feature.f0
        a = psd_feature.S0 * w0 * psd_feature.Q / 4
        b = a / delta
        c = w0 / psd_feature.Q / 2
        d = c * delta
        return [a, b, c, d]
    raise ValueError(f"Feature {type(psd_feature).__name__} not implemented")


def build_approx(J, f0, fM, basis_function="SHO"):
    """src/psd.jl:73-102."""
    sp = f0 * (fM / f0) ** (np.arange(J) / (J - 1))
    if basis_function == "SHO":
        power = 4
    elif basis_function == "DRWCelerite":
        power = 6
    else:
        raise ValueError("Basis function" + basis_function + "not implemented")
    return sp, 1.0 / (1.0 + (sp[:, None] / sp[None, :]) ** power)


def psd_decomp(psd_normalised, spectral_matrix):
    """src/psd.jl:109-112."""
    return np.linalg.solve(spectral_matrix, psd_normalised)


def get_approx_coefficients(psd_model, f0, fM, n_components=20, basis_function="SHO"):
    """src/psd.jl:129-135."""
    sp, Bm = build_approx(n_components, f0, fM, basis_function)
    p = psd_model(sp)
    return psd_decomp(p / p[0], Bm)


def _integral_sho(a, c, x):
    # src/psd.jl:301-305; a may carry leading batch axes
    norm = c * a / (4 * math.sqrt(2))
    poly = (x ** 2 + math.sqrt(2) * c * x + c ** 2) / (x ** 2 - math.sqrt(2) * c * x + c ** 2)
    return np.sum(norm * (np.log(poly) + 2 * np.arctan2(c * math.sqrt(2) * x, (c ** 2 - x ** 2))), axis=-1)


def _integral_drwcelerite(a, c, x):
    # src/psd.jl:318-324
    norm = a * c / 3
    poly = (x ** 2 + math.sqrt(3) * c * x + c ** 2) / (x ** 2 - math.sqrt(3) * c * x + c ** 2)
    cel = 0.5 * np.arctan2(x ** 2 - c ** 2, c * x) + math.sqrt(3) / 4 * np.log(poly)
    return np.sum(norm * (np.arctan(x / c) + cel), axis=-1)


def _integral_celerite(a, b, c, d, x):
    # src/psd.jl:330-334
    num = c ** 2 + (d + 2 * math.pi * x) ** 2
    den = c ** 2 + (d - 2 * math.pi * x) ** 2
    return (2 * a * (np.arctan2(c, d - 2 * math.pi * x) - np.arctan2(c, d + 2 * math.pi * x))
            + b * np.log(num / den)) / (2 * math.pi)


def get_norm_psd(amplitudes, spectral_points, f_min, f_max, basis_function, is_integrated_power, cov_features=None):
    """src/psd.jl:375-395."""
    if is_integrated_power:
        fn = _integral_sho if basis_function == "SHO" else _integral_drwcelerite
        integ = fn(amplitudes, spectral_points, f_max) - fn(amplitudes, spectral_points, f_min)
        if cov_features is not None:
            for a, b, c, d in np.asarray(cov_features).T:
                integ = integ + _integral_celerite(a, b, c, d, f_max) - _integral_celerite(a, b, c, d, f_min)
        return integ
    if basis_function == "SHO":
        return np.sum(amplitudes * spectral_points, axis=-1) * math.pi / math.sqrt(2)
    return np.sum(amplitudes * spectral_points, axis=-1) * 2 * math.pi / 3


def approx(psd_model, f_min, f_max, n_components=20, norm=1.0, S_low=20.0, S_high=20.0, *,
           is_integrated_power=True, basis_function="SHO") -> SumOfCelerite:
    """approx(psd_model, f_min, f_max, n_components, norm, S_low, S_high; is_integrated_power, basis_function)
    src/psd.jl:214-289 — returns the SumOfCelerite covariance."""
    f0 = f_min / S_low
    fM = f_max * S_high
    sp, Bm = build_approx(n_components, f0, fM, basis_function)
    cont, feats = separate_psd(psd_model)
    if cont is None:
        raise AssertionError("The PSD model should contain at least one ContinuumPowerSpectrum component to be approximated")
    p = cont(sp)
    psd_norm = p[0]
    amplitudes = psd_decomp(p / psd_norm, Bm)
    cov_features = None
    if feats is not None:
        cov_features = np.array([convert_feature(f) for f in feats], float).T
        cov_features[0, :] /= psd_norm
        cov_features[1, :] /= psd_norm
    integ = get_norm_psd(amplitudes, sp, f_min, f_max, basis_function, is_integrated_power, cov_features)
    amplitudes = amplitudes * norm / integ
    if cov_features is not None:
        cov_features[0, :] *= norm / integ
        cov_features[1, :] *= norm / integ
    if basis_function == "SHO":
        a = amplitudes * sp * math.pi / math.sqrt(2)
        c = math.sqrt(2) * math.pi * sp
        aa, bb, cc, dd = a, a.copy(), c, c.copy()
    else:
        a = amplitudes * sp * math.pi / 3
        c = math.pi * sp
        aa = np.concatenate([a, a]); bb = np.concatenate([math.sqrt(3) * a, np.zeros(n_components)])
        cc = np.concatenate([c, 2 * c]); dd = np.concatenate([math.sqrt(3) * c, np.zeros(n_components)])
    if cov_features is not None:
        aa = np.concatenate([aa, 2 * cov_features[0, :]]); bb = np.concatenate([bb, 2 * cov_features[1, :]])
        cc = np.concatenate([cc, cov_features[2, :]]); dd = np.concatenate([dd, cov_features[3, :]])
    return SumOfCelerite(aa, bb, cc, dd)


def approx_batch(model, theta, f_min, f_max, n_components=20, norm=1.0, S_low=20.0, S_high=20.0, *,
                 is_integrated_power=True, basis_function="SHO"):
    """Vectorised `approx` for a continuum model class over B parameter rows.

    model: SingleBendingPowerLaw or DoubleBendingPowerLaw (the class); theta: (B, n_psd_params);
    norm: scalar or (B,).  Returns A, Bc of shape (B, J) and the shared c, d of shape (J,)
    (c_j, d_j depend only on the spectral grid, src/psd.jl:250,266-267)."""
    theta = np.atleast_2d(np.asarray(theta))
    # complex parameters are allowed (every operation below is analytic): approx_batch_vjp differentiates by the complex step
    cplx = np.iscomplexobj(theta) or np.iscomplexobj(norm)
    theta = theta.astype(complex if cplx else float)
    B = theta.shape[0]
    f0 = f_min / S_low
    fM = f_max * S_high
    sp, Bm = build_approx(n_components, f0, fM, basis_function)
    psd = np.stack([model(*row)(sp) for row in theta]) if (B < 64 and not cplx) else _eval_model_batch(model, theta, sp)
    psd = psd / psd[:, :1]
    amplitudes = np.ascontiguousarray(np.linalg.solve(Bm, psd.T).T)  # one LU, B right-hand sides; C order (B, J)
    integ = get_norm_psd(amplitudes, sp, f_min, f_max, basis_function, is_integrated_power)
    amplitudes = amplitudes * (np.broadcast_to(np.asarray(norm, complex if cplx else float), (B,)) / integ)[:, None]
    if basis_function == "SHO":
        a = amplitudes * sp * math.pi / math.sqrt(2)
        c = math.sqrt(2) * math.pi * sp
        return a, a.copy(), c, c.copy()
    a = amplitudes * sp * math.pi / 3
    c = math.pi * sp
    A = np.concatenate([a, a], axis=1)
    Bc = np.concatenate([math.sqrt(3) * a, np.zeros_like(a)], axis=1)
    return A, Bc, np.concatenate([c, 2 * c]), np.concatenate([math.sqrt(3) * c, np.zeros(n_components)])


def approx_batch_vjp(model, theta, f_min, f_max, n_components, norm, grad_a, grad_b, S_low=20.0, S_high=20.0, *,
                     is_integrated_power=True, basis_function="SHO", h=1e-30):
    """Chain rule through `approx`: given dL/da, dL/db (B, J) returns dL/dtheta (B, P) and dL/dnorm (B,).
    The Jacobian-vector products come from one complex step per parameter (exact to rounding; `approx` is a chain of
    analytic operations: power laws, one linear solve, a normalising ratio)."""
    theta = np.atleast_2d(np.asarray(theta, float))
    B, P = theta.shape
    norm = np.broadcast_to(np.asarray(norm, float), (B,))
    if model not in (SingleBendingPowerLaw, DoubleBendingPowerLaw):
        raise ValueError("approx_batch_vjp supports SingleBendingPowerLaw and DoubleBendingPowerLaw")
    gth = np.empty((B, P))
    for k in range(P):
        th = theta.astype(complex)
        th[:, k] += 1j * h
        Ak, Bk, _, _ = approx_batch(model, th, f_min, f_max, n_components, norm, S_low, S_high,
                                    is_integrated_power=is_integrated_power, basis_function=basis_function)
        gth[:, k] = (grad_a * Ak.imag + grad_b * Bk.imag).sum(axis=1) / h
    A, Bc, _, _ = approx_batch(model, theta, f_min, f_max, n_components, norm, S_low, S_high,
                               is_integrated_power=is_integrated_power, basis_function=basis_function)
    gnorm = (grad_a * A + grad_b * Bc).sum(axis=1) / norm      # (a, b) are proportional to norm
    return gth, gnorm


def _eval_model_batch(model, theta, sp):
    f = sp[None, :]
    if model is SingleBendingPowerLaw:
        a1, f1, a2 = (theta[:, i:i + 1] for i in range(3))
        return (f / f1) ** (-a1) / (1 + (f / f1) ** (a2 - a1))
    if model is DoubleBendingPowerLaw:
        a1, f1, a2, f2, a3 = (theta[:, i:i + 1] for i in range(5))
        return (f / f1) ** (-a1) / (1 + (f / f1) ** (a2 - a1)) / (1 + (f / f2) ** (a3 - a2))
    return np.stack([model(*row)(sp) for row in theta])
