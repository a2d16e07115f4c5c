"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement of the reference's (mlefkir/Pioran.jl v1.2.0) ScalableGP log-likelihood path:
  * ctypes front-end to oracle/celerite_oracle.c (celerite logl, dense NLL, sim);
  * a numpy twin of logl (independent second implementation, forward-only form);
  * numpy restatements of the host-side coefficient sources the reference feeds the solver with
    (approx / PSD models / SHO / Exp / Celerite / QPO / CARMA -> (a, b, c, d)).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (pioran.jl_amd/) must never import it.

Pinning status: PINNED against outputs of the reference itself — see celerite_oracle.c header and
tests/test_oracle.py.  Citations are file:line relative to the reference root.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"
_SRC = _HERE / "celerite_oracle.c"
_LIB = _BUILD / "liboracle.so"
# no FMA contraction: Julia does not fuse a*b+c either, keep the reference's rounding
_CFLAGS = ["-O3", "-march=x86-64-v3", "-mtune=native", "-ffp-contract=off", "-fopenmp",
           "-fPIC", "-shared", "-std=c11", "-D_GNU_SOURCE"]

_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (seconds)."""
    srcs = [_SRC, _SRC.with_name("celerite_oracle_cstep.c")]
    if not force and _LIB.exists() and all(_LIB.stat().st_mtime >= f.stat().st_mtime for f in srcs):
        return _LIB
    _BUILD.mkdir(exist_ok=True)
    tmp = _BUILD / f"liboracle.{os.getpid()}.so"
    subprocess.run(["gcc", *_CFLAGS, *map(str, srcs), "-o", str(tmp), "-lm"], check=True)
    os.replace(tmp, _LIB)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            L = ctypes.CDLL(str(_LIB))
        except OSError:
            build(force=True)
            L = ctypes.CDLL(str(_LIB))
        i64 = ctypes.c_int64
        L.oracle_logl.restype = ctypes.c_double
        L.oracle_logl.argtypes = [i64, i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]
        L.oracle_logl_detail.restype = ctypes.c_double
        L.oracle_logl_detail.argtypes = [i64, i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_logl_batch.restype = None
        L.oracle_logl_batch.argtypes = [i64, i64, i64, _dp, _dp, _dp, _dp, ctypes.c_int, _dp, _dp,
                                        _dp, _dp, _dp, _dp, _ip, ctypes.c_int]
        L.oracle_dense_nll.restype = ctypes.c_double
        L.oracle_dense_nll.argtypes = [i64, i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.oracle_kappa.restype = ctypes.c_double
        L.oracle_kappa.argtypes = [i64, _dp, _dp, _dp, _dp, ctypes.c_double]
        L.oracle_logl_complex.restype = ctypes.c_double
        L.oracle_logl_complex.argtypes = [i64, i64] + [_dp] * 11 + [_dp]
        L.oracle_logl_complex_cd.restype = ctypes.c_double
        L.oracle_logl_complex_cd.argtypes = [i64, i64] + [_dp] * 13 + [_dp]
        L.oracle_predict.restype = None
        L.oracle_predict.argtypes = [i64, i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, i64, _dp, _dp]
        L.oracle_sim.restype = None
        L.oracle_sim.argtypes = [i64, i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _c(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return x, x.ctypes.data_as(_dp)


# ---------------------------------------------------------------------------------------------
# higher-precision truth: the same recurrence in __float128 (celerite_oracle_q.c, libquadmath) — its own small library, built on first use
# ---------------------------------------------------------------------------------------------
_SRC_Q = _HERE / "celerite_oracle_q.c"
_LIB_Q = _BUILD / "liboracle_q.so"
_lib_q = None


def build_quad(force: bool = False) -> Path:
    if not force and _LIB_Q.exists() and _LIB_Q.stat().st_mtime >= _SRC_Q.stat().st_mtime:
        return _LIB_Q
    _BUILD.mkdir(exist_ok=True)
    tmp = _BUILD / f"liboracle_q.{os.getpid()}.so"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", str(_SRC_Q), "-o", str(tmp), "-lquadmath", "-lm"], check=True)
    os.replace(tmp, _LIB_Q)
    return _LIB_Q


def _libq():
    global _lib_q
    if _lib_q is None:
        build_quad()
        L = ctypes.CDLL(str(_LIB_Q))
        i64 = ctypes.c_int64
        L.oracle_logl_quad.restype = ctypes.c_double
        L.oracle_logl_quad.argtypes = [i64, i64] + [_dp] * 7 + [_dp]
        L.oracle_logl_quad_batch.restype = None
        L.oracle_logl_quad_batch.argtypes = [i64, i64, i64] + [_dp] * 11 + [ctypes.c_int]
        _lib_q = L
    return _lib_q


def logl_quad(a, b, c, d, t, y, sigma2, return_dmin=False):
    """logl (src/celerite_solver.jl:312-334) evaluated in __float128 and rounded to fp64: the truth the fp64 evaluations are held against."""
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    dm = ctypes.c_double(0.0)
    v = _libq().oracle_logl_quad(len(t), len(a), pa, pb, pc, pd, pt, py, ps, ctypes.cast(ctypes.byref(dm), _dp))
    return (v, dm.value) if return_dmin else v


def logl_quad_batch(A, Bc, C, Dd, t, y, sigma2, mu=None, nu=None, nthreads=1, return_dmin=False):
    """B independent quad-precision evaluations (shared c, d); mu / nu applied in fp64 exactly as the fp64 paths apply them."""
    A, pA = _c(A); Bc, pB = _c(Bc); C, pC = _c(C); Dd, pD = _c(Dd)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    B, J = A.shape
    assert C.ndim == 1
    pmu = pnu = None
    if mu is not None:
        mu, pmu = _c(mu)
    if nu is not None:
        nu, pnu = _c(nu)
    out = np.empty(B); dmin = np.empty(B)
    _libq().oracle_logl_quad_batch(len(t), J, B, pA, pB, pC, pD, pmu, pnu, pt, py, ps, out.ctypes.data_as(_dp), dmin.ctypes.data_as(_dp),
                                   int(nthreads))
    return (out, dmin) if return_dmin else out


# ---------------------------------------------------------------------------------------------
# solver front-ends (C)
# ---------------------------------------------------------------------------------------------
def logl(a, b, c, d, t, y, sigma2, return_status=False):
    """src/celerite_solver.jl:312-334 logl(a, b, c, d, tau, y, sigma2)."""
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    st = ctypes.c_int32(0)
    v = lib().oracle_logl(len(t), len(a), pa, pb, pc, pd, pt, py, ps, ctypes.byref(st))
    return (v, st.value) if return_status else v


def logl_detail(a, b, c, d, t, y, sigma2):
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    D = np.empty(len(t)); z = np.empty(len(t))
    v = lib().oracle_logl_detail(len(t), len(a), pa, pb, pc, pd, pt, py, ps,
                                 D.ctypes.data_as(_dp), z.ctypes.data_as(_dp))
    return v, D, z


def logl_batch(A, Bc, C, Dd, t, y, sigma2, mu=None, nu=None, nthreads=1, return_status=False):
    """B independent logl calls. A, Bc: (B, J) arrays (row b = draw b). C, Dd: (J,) shared or (B, J)."""
    A, pA = _c(A); Bc, pB = _c(Bc); C, pC = _c(C); Dd, pD = _c(Dd)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    B, J = A.shape
    cd_shared = int(C.ndim == 1)
    pmu = pnu = None
    if mu is not None:
        mu, pmu = _c(mu)
    if nu is not None:
        nu, pnu = _c(nu)
    out = np.empty(B); st = np.zeros(B, dtype=np.int32)
    lib().oracle_logl_batch(len(t), J, B, pA, pB, pC, pD, cd_shared, pmu, pnu, pt, py, ps,
                            out.ctypes.data_as(_dp), st.ctypes.data_as(_ip), int(nthreads))
    return (out, st) if return_status else out


def dense_nll(a, b, c, d, t, y, sigma2):
    """src/direct_solver.jl:6-21 log_likelihood_direct — returns the POSITIVE NLL like the reference."""
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    t, pt = _c(t); y, py = _c(y); s2, ps = _c(sigma2)
    return lib().oracle_dense_nll(len(t), len(a), pa, pb, pc, pd, pt, py, ps)


def kappa(a, b, c, d, tau):
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    return lib().oracle_kappa(len(a), pa, pb, pc, pd, float(tau))


def sim(a, b, c, d, t, sigma2, q):
    """src/celerite_solver.jl:515-549 sim with the normal deviates q supplied by the caller."""
    a, pa = _c(a); b, pb = _c(b); c, pc = _c(c); d, pd = _c(d)
    t, pt = _c(t); s2, ps = _c(sigma2); q, pq = _c(q)
    out = np.empty(len(t))
    lib().oracle_sim(len(t), len(a), pa, pb, pc, pd, pt, ps, pq, out.ctypes.data_as(_dp))
    return out


# ---------------------------------------------------------------------------------------------
# numpy twin of logl (independent; forward-only quadratic form)
# ---------------------------------------------------------------------------------------------
def logl_numpy(a, b, c, d, t, y, sigma2):
    a = np.asarray(a, float); b = np.asarray(b, float)
    c = np.asarray(c, float); d = np.asarray(d, float)
    t = np.asarray(t, float); y = np.asarray(y, float); s2 = np.asarray(sigma2, float)
    J = len(a); R = 2 * J; N = len(t)
    al = np.empty(R); be = np.empty(R)
    al[0::2] = a; be[0::2] = b; al[1::2] = -b; be[1::2] = a
    cc = np.repeat(c, 2)
    suma = a.sum()
    S = np.zeros((R, R))
    D = suma + s2[0]
    v = np.empty(R)
    v[0::2] = np.cos(d * t[0]); v[1::2] = np.sin(d * t[0])
    w = v / D
    f = np.zeros(R)
    z = y[0]
    ld = np.log(D)
    q2 = z * z / D
    for n in range(1, N):
        co = np.cos(d * t[n]); si = np.sin(d * t[n])
        ph = np.exp(-cc * (t[n] - t[n - 1]))
        u = al * np.repeat(co, 2) + be * np.repeat(si, 2)
        v[0::2] = co; v[1::2] = si
        S = np.outer(ph, ph) * (S + D * np.outer(w, w))
        f = ph * (f + w * z)
        q = S @ u
        D = suma + s2[n] - u @ q
        w = (v - q) / D
        z = y[n] - u @ f
        ld += np.log(abs(D))
        q2 += z * z / D
    return -0.5 * ld - 0.5 * N * np.log(2 * np.pi) - 0.5 * q2


def logl_grad(a, b, c, d, t, y, sigma2, series=False, h=1e-30, cd=False):
    """Exact derivatives of logl (src/celerite_solver.jl:312-334) by the complex step on the complex twin of the C
    restatement: d/da_j, d/db_j (J each), with cd=True d/dc_j, d/dd_j and, with series=True, d/dy_n, d/dsigma2_n (N each;
    2N more evaluations)."""
    a, b, c, d, t, y, sigma2 = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b, c, d, t, y, sigma2))
    J, N = len(a), len(t)
    P = lambda v: None if v is None else v.ctypes.data_as(_dp)
    im = ctypes.c_double()
    def run(a_im=None, b_im=None, y_im=None, s_im=None, c_im=None, d_im=None):
        lib().oracle_logl_complex_cd(N, J, P(a), P(a_im), P(b), P(b_im), P(c), P(c_im), P(d), P(d_im), P(t), P(y), P(y_im),
                                     P(sigma2), P(s_im), ctypes.byref(im))
        return im.value / h
    def sweep(n, key):
        out = np.empty(n)
        for k in range(n):
            e = np.zeros(n); e[k] = h
            out[k] = run(**{key: e})
        return out
    res = {"grad_a": sweep(J, "a_im"), "grad_b": sweep(J, "b_im")}
    if cd:
        res["grad_c"] = sweep(J, "c_im"); res["grad_d"] = sweep(J, "d_im")
    if series:
        res["grad_y"] = sweep(N, "y_im"); res["grad_sigma2"] = sweep(N, "s_im")
    return res


def logl_dir(a, b, c, d, t, y, sigma2, da=None, db=None, dy=None, ds2=None, h=1e-30):
    """Directional derivative of logl along (da, db, dy, ds2) by one complex step (exact to rounding)."""
    a, b, c, d, t, y, sigma2 = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b, c, d, t, y, sigma2))
    P = lambda v: None if v is None else np.ascontiguousarray(h * np.asarray(v, dtype=np.float64))
    parts = [P(da), P(db), P(dy), P(ds2)]
    Q = lambda v: None if v is None else v.ctypes.data_as(_dp)
    im = ctypes.c_double()
    lib().oracle_logl_complex(len(t), len(a), Q(a), Q(parts[0]), Q(b), Q(parts[1]), Q(c), Q(d), Q(t), Q(y), Q(parts[2]),
                              Q(sigma2), Q(parts[3]), ctypes.byref(im))
    return im.value / h


def predict(a, b, c, d, tau, t, y, sigma2):
    """pred(a, b, c, d, tau, t, y, sigma2)   src/celerite_solver.jl:363-483 (tau ascending); C restatement."""
    a, b, c, d, tau, t, y, sigma2 = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b, c, d, tau, t, y, sigma2))
    out = np.empty(len(tau))
    P = lambda v: v.ctypes.data_as(_dp)
    lib().oracle_predict(len(t), len(a), P(a), P(b), P(c), P(d), P(t), P(y), P(sigma2), len(tau), P(tau), P(out))
    return out


def predict_direct_numpy(a, b, c, d, tau, t, y, sigma2):
    """predict_direct (src/direct_solver.jl:75-119): K_tau0 K0^-1 y with dense matrices and a Cholesky solve."""
    a, b, c, d, tau, t, y, sigma2 = (np.asarray(v, dtype=np.float64) for v in (a, b, c, d, tau, t, y, sigma2))
    def kern(dt):
        dt = np.abs(dt)[..., None]
        return (np.exp(-c * dt) * (a * np.cos(d * dt) + b * np.sin(d * dt))).sum(-1)
    K0 = kern(t[:, None] - t[None, :]) + np.diag(sigma2)
    Kt0 = kern(tau[:, None] - t[None, :])
    L = np.linalg.cholesky(K0)
    z = np.linalg.solve(L.T, np.linalg.solve(L, y))
    return Kt0 @ z


def predict_cov_numpy(a, b, c, d, tau, t, sigma2):
    """predict_cov (src/direct_solver.jl:28-69): K_tau - w'w, w = L \\ K_tau0' with L the Cholesky factor of K0 + diag(sigma2)."""
    a, b, c, d, tau, t, sigma2 = (np.asarray(v, dtype=np.float64) for v in (a, b, c, d, tau, t, sigma2))
    def kern(dt):
        dt = np.abs(dt)[..., None]
        return (np.exp(-c * dt) * (a * np.cos(d * dt) + b * np.sin(d * dt))).sum(-1)
    K0 = kern(t[:, None] - t[None, :]) + np.diag(sigma2)
    Kt0 = kern(tau[:, None] - t[None, :])
    Kt = kern(tau[:, None] - tau[None, :])
    L = np.linalg.cholesky(K0)
    w = np.linalg.solve(L, Kt0.T)
    return Kt - w.T @ w


def dense_nll_numpy(a, b, c, d, t, y, sigma2):
    """src/direct_solver.jl:6-21 with numpy's LAPACK Cholesky (second implementation)."""
    a = np.asarray(a, float); b = np.asarray(b, float)
    c = np.asarray(c, float); d = np.asarray(d, float)
    t = np.asarray(t, float); y = np.asarray(y, float)
    tau = np.abs(t[:, None] - t[None, :])
    K = np.zeros_like(tau)
    for j in range(len(a)):
        K += np.exp(-c[j] * tau) * (a[j] * np.cos(d[j] * tau) + b[j] * np.sin(d[j] * tau))
    K = K + np.diag(np.asarray(sigma2, float))
    L = np.linalg.cholesky(K)
    import scipy.linalg
    z = scipy.linalg.solve_triangular(L, y, lower=True)
    return np.log(np.diag(L)).sum() + 0.5 * z @ z + 0.5 * len(t) * np.log(2 * np.pi)


# ---------------------------------------------------------------------------------------------
# coefficient sources
# ---------------------------------------------------------------------------------------------
def celerite_coefs_celerite(a, b, c, d):
    """src/Celerite.jl:33-39."""
    return [a, b, c, d]


def celerite_coefs_sho(A, w0, Q):
    """src/SHO.jl:31-41 (only Q == 1/sqrt(2))."""
    if Q == 1 / np.sqrt(2):
        cc = np.sqrt(2) / 2 * w0
        return [A, A, cc, cc]
    raise ValueError("SHO with Q≠1/√2 not implemented yet")


def celerite_coefs_exp(A, alpha):
    """src/Exp.jl:29-33."""
    return [A / 2, 0.0, alpha, 0.0]


def convert_feature_qpo(S0, f0, Q):
    """src/psd.jl:15-27 convert_feature(::QPO)."""
    delta = np.sqrt(4 * Q ** 2 - 1)
    w0 = 2 * np.pi * f0
    a = S0 * w0 * Q / 4
    b = a / delta
    c = w0 / Q / 2
    d = c * delta
    return [a, b, c, d]


def carma_celerite_coefs(p, r_alpha, beta, norm, is_integrated_power=True):
    """src/CARMA.jl:98-143 CARMA_celerite_coefs."""
    r_alpha = np.asarray(r_alpha, dtype=complex)
    beta = np.asarray(beta, dtype=float)
    J = p // 2 if p % 2 == 0 else (p - 1) // 2 + 1
    a = np.empty(J); b = np.empty(J); c = np.empty(J); d = np.empty(J)
    for k, rk in enumerate(r_alpha[0::2]):
        num_1 = 0.0; num_2 = 0.0
        for l, bl in enumerate(beta):
            num_1 = num_1 + bl * rk ** l
            num_2 = num_2 + bl * (-rk) ** l
        frac = -num_1 * num_2 / rk.real
        for rj in r_alpha:
            if rj != rk:
                frac = frac / ((rj - rk) * (np.conj(rj) + rk))
        if (k + 1) != J or p % 2 == 0:
            a[k] = 2 * frac.real; b[k] = 2 * frac.imag; c[k] = -rk.real; d[k] = -rk.imag
        else:
            a[k] = frac.real; b[k] = 0.0; c[k] = -rk.real; d[k] = 0.0
    va = norm
    if is_integrated_power:
        va = va / a.sum()
    return a * va, b * va, c, d


# PSD models — Tonari.jl ^0.2 (un-vendored dependency, Project.toml:46); closed forms pinned by the
# reference's test/test_psd.jl:3-13.
def single_bending_power_law(f, a1, f1, a2):
    f = np.asarray(f, float)
    return (f / f1) ** (-a1) / (1 + (f / f1) ** (a2 - a1))


def double_bending_power_law(f, a1, f1, a2, f2, a3):
    f = np.asarray(f, float)
    return (f / f1) ** (-a1) / (1 + (f / f1) ** (a2 - a1)) / (1 + (f / f2) ** (a3 - a2))


def build_approx(J, f0, fM, basis_function="SHO"):
    """src/psd.jl:73-102 build_approx / init_psd_decomp!."""
    sp = np.array([f0 * (fM / f0) ** (j / (J - 1)) for j in range(J)])
    p = {"SHO": 4, "DRWCelerite": 6}[basis_function]
    B = 1.0 / (1.0 + (sp[:, None] / sp[None, :]) ** p)
    return sp, B


def psd_decomp(psd_normalised, spectral_matrix):
    """src/psd.jl:109-112 (Julia `\\` on a square matrix = LU with partial pivoting)."""
    return np.linalg.solve(spectral_matrix, psd_normalised)


def get_approx_coefficients(psd, f0, fM, n_components=20, basis_function="SHO"):
    """src/psd.jl:129-135."""
    sp, B = build_approx(n_components, f0, fM, basis_function)
    p = psd(sp)
    return psd_decomp(p / p[0], B)


def integral_sho(a, c, x):
    """src/psd.jl:301-305."""
    norm = c * a / (4 * np.sqrt(2))
    poly = (x ** 2 + np.sqrt(2) * c * x + c ** 2) / (x ** 2 - np.sqrt(2) * c * x + c ** 2)
    return np.sum(norm * (np.log(poly) + 2 * np.arctan2(c * np.sqrt(2) * x, (c ** 2 - x ** 2))))


def integral_drwcelerite(a, c, x):
    """src/psd.jl:318-324."""
    norm = a * c / 3
    drw = np.arctan(x / c)
    poly = (x ** 2 + np.sqrt(3) * c * x + c ** 2) / (x ** 2 - np.sqrt(3) * c * x + c ** 2)
    cel = 0.5 * np.arctan2(x ** 2 - c ** 2, c * x) + np.sqrt(3) / 4 * np.log(poly)
    return np.sum(norm * (drw + cel))


def integral_celerite(a, b, c, d, x):
    """src/psd.jl:330-334."""
    num = c ** 2 + (d + 2 * np.pi * x) ** 2
    den = c ** 2 + (d - 2 * np.pi * x) ** 2
    return (2 * a * (np.arctan2(c, d - 2 * np.pi * x) - np.arctan2(c, d + 2 * np.pi * x))
            + b * np.log(num / den)) / (2 * np.pi)


def get_norm_psd(amplitudes, sp, f_min, f_max, basis_function, is_integrated_power, cov_features=None):
    """src/psd.jl:375-395."""
    if is_integrated_power:
        fn = {"SHO": integral_sho, "DRWCelerite": integral_drwcelerite}[basis_function]
        integ = fn(amplitudes, sp, f_max) - fn(amplitudes, sp, f_min)
        if cov_features is not None:
            for col in np.asarray(cov_features).T:
                a, b, c, d = col
                integ += integral_celerite(a, b, c, d, f_max) - integral_celerite(a, b, c, d, f_min)
        return integ
    if basis_function == "SHO":
        return np.sum(amplitudes * sp) * np.pi / np.sqrt(2)
    return np.sum(amplitudes * sp) * 2 * np.pi / 3


def approx(psd, f_min, f_max, n_components=20, norm=1.0, S_low=20.0, S_high=20.0,
           is_integrated_power=True, basis_function="SHO", qpo_features=None):
    """src/psd.jl:214-289 approx.  `psd` is the continuum callable; `qpo_features` an optional list of
    (S0, f0, Q) triples (Tonari's QPO).  Returns (a, b, c, d)."""
    f0 = f_min / S_low
    fM = f_max * S_high
    sp, B = build_approx(n_components, f0, fM, basis_function)
    p = psd(sp)
    psd_norm = p[0]
    amplitudes = psd_decomp(p / psd_norm, B)
    cov_features = None
    if qpo_features:
        cov_features = np.array([convert_feature_qpo(*q) for q in qpo_features], float).T  # 4 x nq
        cov_features[0, :] /= psd_norm
        cov_features[1, :] /= psd_norm
    integ = get_norm_psd(amplitudes, sp, f_min, f_max, basis_function, is_integrated_power, cov_features)
    amplitudes = amplitudes * norm / integ
    if cov_features is not None:
        cov_features[0, :] *= norm / integ
        cov_features[1, :] *= norm / integ
    if basis_function == "SHO":
        a = amplitudes * sp * np.pi / np.sqrt(2)
        c = np.sqrt(2) * np.pi * sp
        aa, bb, cc, dd = a, a.copy(), c, c.copy()
    elif basis_function == "DRWCelerite":
        a = amplitudes * sp * np.pi / 3
        b = np.sqrt(3) * a
        c = np.pi * sp
        d = np.sqrt(3) * c
        aa = np.concatenate([a, a]); bb = np.concatenate([b, np.zeros(n_components)])
        cc = np.concatenate([c, 2 * c]); dd = np.concatenate([d, np.zeros(n_components)])
    else:
        raise ValueError("Basis function" + basis_function + "not implemented")
    if cov_features is not None:
        aa = np.concatenate([aa, 2 * cov_features[0, :]]); bb = np.concatenate([bb, 2 * cov_features[1, :]])
        cc = np.concatenate([cc, cov_features[2, :]]); dd = np.concatenate([dd, cov_features[3, :]])
    return aa, bb, cc, dd


# ---------------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md section 8(d) (replaces the missing benchmark/simulate_long.txt)
# ---------------------------------------------------------------------------------------------
def synthetic_series(N=10_000, seed=1234):
    """Irregular series: gaps 0.05 + Exp(0.95); yerr ~ U(0.007, 0.05); y = exact GP draw (oracle sim)
    from SingleBendingPowerLaw(0.82, 0.01, 3.3), SHO-20, variance 1 (benchmark/benchmarks.jl:37) + noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    gaps = 0.05 + rng.exponential(0.95, size=N)
    t = np.cumsum(gaps) - gaps[0]
    yerr = rng.uniform(0.007, 0.05, size=N)
    f_min = 1.0 / (t[-1] - t[0]); f_max = 1.0 / (2 * np.min(np.diff(t)))
    a, b, c, d = approx(lambda f: single_bending_power_law(f, 0.82, 0.01, 3.3), f_min, f_max, 20, 1.0)
    q = rng.standard_normal(N)
    y = sim(a, b, c, d, t, np.zeros(N), q) + yerr * rng.standard_normal(N)
    return t, y, yerr


def synthetic_theta(B, t, y, seed=4321):
    """Parameter draws from the priors of benchmark/benchmarks.jl:51-56 (mu Gaussian)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f_min = 1.0 / (t[-1] - t[0]); f_max = 1.0 / (2 * np.min(np.diff(t)))
    th = np.empty((B, 6))
    th[:, 0] = rng.uniform(-0.25, 2.0, B)
    th[:, 1] = np.exp(rng.uniform(np.log(f_min), np.log(f_max), B))
    th[:, 2] = rng.uniform(1.5, 4.0, B)
    th[:, 3] = np.exp(np.log(0.5) + 1.25 * rng.standard_normal(B))
    th[:, 4] = rng.gamma(2.0, 0.5, B)
    th[:, 5] = np.mean(y) + np.std(y) * rng.standard_normal(B)
    return th


def theta_to_coefs(theta, t, n_components=20, basis_function="SHO", is_integrated_power=True):
    """theta[B,6] = (alpha1, f1, alpha2, variance, nu, mu) -> A,Bc (B,J), C,Dd (J,), mu, nu."""
    f_min = 1.0 / (t[-1] - t[0]); f_max = 1.0 / (2 * np.min(np.diff(t)))
    A = []; Bc = []; C = Dd = None
    for th in theta:
        a, b, c, d = approx(lambda f: single_bending_power_law(f, th[0], th[1], th[2]),
                            f_min, f_max, n_components, th[3], basis_function=basis_function,
                            is_integrated_power=is_integrated_power)
        A.append(a); Bc.append(b); C = c; Dd = d
    return np.array(A), np.array(Bc), C, Dd, theta[:, 5].copy(), theta[:, 4].copy()
