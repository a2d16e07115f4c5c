"""GP facade + solver entry points of the hot path, host side (Python over the C ABI).

Mirrors, with the same names and argument meaning:
  ScalableGP(mu, kernel[, solver]) / f(t, sigma2) / logpdf(fx, y)   src/scalable_GP.jl:24-42,162-166
  log_likelihood(cov, tau, y, sigma2; solver)                       src/celerite_solver.jl:262-294
  logl(a, b, c, d, tau, y, sigma2)                                  src/celerite_solver.jl:312-334
  log_likelihood_direct(cov, t, y, sigma2)  (returns +NLL)          src/direct_solver.jl:6-21
plus the batched entry the reference lacks: logpdf_batch / Dataset.logl_batch.

Everything numerical happens in libpioran_hip.so on the GPU.  No CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .kernels import SemiSeparable, SumOfCelerite

_SOLVERS = ("celerite", "celerite_matrix")


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _ptr(x):
    return None if x is None else ctypes.c_void_p(x.ctypes.data)


class Context:
    """One GPU + one HIP stream (pioran_ctx).  One per host thread / rank."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = _lib.lib()
        h = ctypes.c_void_p()
        if stream is None:
            _lib.check(L.pioran_ctx_create(int(device), ctypes.byref(h)))
        else:
            _lib.check(L.pioran_ctx_create_on_stream(int(device), ctypes.c_void_p(stream), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if self._h:
            _lib.lib().pioran_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(_lib.lib().pioran_ctx_synchronize(self._h), self._h)

    def trim(self):
        """Release the context's scratch buffers (gradient / prediction workspaces, staging, the dense slab)."""
        _lib.check(_lib.lib().pioran_ctx_trim(self._h), self._h)

    def set_option(self, key: str, value=None):
        """Diagnostic switch of this context (pioran_ctx_set_option): "scan_config" (a configuration name, "wide", "block" or
        None), "no_wide" / "no_block" / "no_paired" / "no_mixed" / "force_fallback" / "win2" / "no_win2" (truthy = on).
        Tests and tuning tools only."""
        if value is None or value is False:
            v = None
        elif value is True:
            v = b"1"
        else:
            v = str(value).encode()
        _lib.check(_lib.lib().pioran_ctx_set_option(self._h, key.encode(), v), self._h)

    def event_record(self, slot: int):
        _lib.check(_lib.lib().pioran_ctx_event_record(self._h, slot), self._h)

    def fp64_probe(self, waves_per_simd: int = 2, ms: float = 10.0) -> float:
        """The FP64 FMA rate (TFLOP/s) this device sustains now at `waves_per_simd` wavefronts per SIMD (pioran_ctx_fp64_probe)."""
        out = ctypes.c_double()
        _lib.check(_lib.lib().pioran_ctx_fp64_probe(self._h, int(waves_per_simd), float(ms), ctypes.byref(out)), self._h)
        return out.value

    def event_elapsed_ms(self, a: int, b: int) -> float:
        ms = ctypes.c_float()
        _lib.check(_lib.lib().pioran_ctx_event_elapsed_ms(self._h, a, b, ctypes.byref(ms)), self._h)
        return ms.value

    # -- scalar drop-ins ---------------------------------------------------------------------
    def logl(self, a, b, c, d, tau, y, sigma2, return_status=False):
        a, b, c, d, tau, y, sigma2 = map(_f64, (a, b, c, d, tau, y, sigma2))
        if not (len(a) == len(b) == len(c) == len(d)) or not (len(tau) == len(y) == len(sigma2)):
            raise ValueError("inconsistent lengths")
        out = ctypes.c_double()
        st = ctypes.c_int32()
        _lib.check(_lib.lib().pioran_celerite_logl(self._h, len(tau), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d),
                                                   _ptr(tau), _ptr(y), _ptr(sigma2), ctypes.byref(out),
                                                   ctypes.byref(st)), self._h)
        return (out.value, st.value) if return_status else out.value

    def simulate(self, A, Bc, C, Dd, t, sigma2, q):
        """GP realisations from standard-normal draws q (B, N): sim of src/celerite_solver.jl:515-549 for B coefficient
        sets (A, Bc: (B, J)) sharing (C, Dd: (J,)).  Returns (B, N)."""
        A, Bc, C, Dd, t, sigma2, q = map(_f64, (A, Bc, C, Dd, t, sigma2, q))
        if A.ndim != 2 or A.shape != Bc.shape or C.shape not in ((A.shape[1],), A.shape) or Dd.shape != C.shape:
            raise ValueError("A, Bc must be (B, J) and C, Dd (J,) or (B, J)")
        B, J = A.shape
        N = len(t)
        if sigma2.shape != (N,) or q.shape != (B, N):
            raise ValueError("sigma2 must be (N,) and q (B, N)")
        out = np.empty((B, N))
        _lib.check(_lib.lib().pioran_celerite_simulate(self._h, N, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(C.ndim == 1), _ptr(t),
                                                       _ptr(sigma2), _ptr(q), _ptr(out)), self._h)
        return out

    def dense_nll(self, a, b, c, d, t, y, sigma2, return_info=False):
        a, b, c, d, t, y, sigma2 = map(_f64, (a, b, c, d, t, y, sigma2))
        out = ctypes.c_double()
        info = ctypes.c_int32()
        _lib.check(_lib.lib().pioran_dense_nll(self._h, len(t), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(t),
                                               _ptr(y), _ptr(sigma2), ctypes.byref(out), ctypes.byref(info)), self._h)
        return (out.value, info.value) if return_info else out.value

    def dense_nll_batch(self, A, Bc, C, Dd, t, y, sigma2, mu=None, nu=None, return_info=False):
        """B dense +NLL values on one data set (pioran_dense_nll_batch): A, Bc (B, J); C, Dd (J,) or (B, J); mu, nu (B,) or
        None.  Independent factorisations run concurrently on the device."""
        A, Bc, C, Dd, t, y, sigma2 = map(_f64, (A, Bc, C, Dd, t, y, sigma2))
        if A.ndim != 2 or A.shape != Bc.shape:
            raise ValueError("A, Bc must be (B, J)")
        B, J = A.shape
        cd_shared = C.ndim == 1
        if C.shape != Dd.shape or C.shape != ((J,) if cd_shared else (B, J)):
            raise ValueError("C, Dd must be (J,) or (B, J)")
        mu = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        out = np.empty(B); info = np.zeros(B, dtype=np.int32)
        _lib.check(_lib.lib().pioran_dense_nll_batch(self._h, len(t), J, B, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(cd_shared),
                                                     _ptr(t), _ptr(y), _ptr(sigma2), _ptr(mu), _ptr(nu), _ptr(out), _ptr(info)), self._h)
        return (out, info) if return_info else out

    def dense_nll_timed(self, a, b, c, d, t, y, sigma2):
        """dense_nll plus the event-timed phases of the call: (nll, info, {"build_ms", "factor_ms", "finish_ms"})."""
        a, b, c, d, t, y, sigma2 = map(_f64, (a, b, c, d, t, y, sigma2))
        out = ctypes.c_double()
        info = ctypes.c_int32()
        ms = (ctypes.c_float * 3)()
        _lib.check(_lib.lib().pioran_dense_nll_timed(self._h, len(t), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(t),
                                                     _ptr(y), _ptr(sigma2), ctypes.byref(out), ctypes.byref(info), ms), self._h)
        return out.value, info.value, {"build_ms": ms[0], "factor_ms": ms[1], "finish_ms": ms[2]}

    def dense_predict_cov(self, a, b, c, d, tau, t, sigma2, return_info=False):
        """predict_cov(cov, tau, t, sigma2): posterior covariance at tau, (M, M)   src/direct_solver.jl:28-69."""
        a, b, c, d, tau, t, sigma2 = map(_f64, (a, b, c, d, tau, t, sigma2))
        out = np.empty((len(tau), len(tau)))
        info = ctypes.c_int32()
        _lib.check(_lib.lib().pioran_dense_predict_cov(self._h, len(t), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(t),
                                                       _ptr(sigma2), len(tau), _ptr(tau), _ptr(out), ctypes.byref(info)),
                   self._h)
        return (out, info.value) if return_info else out

    def dense_predict(self, a, b, c, d, tau, t, y, sigma2, with_covariance=False, return_info=False):
        """predict_direct(cov, tau, t, y, sigma2[, with_covariance])   src/direct_solver.jl:75-119 (dense solver)."""
        a, b, c, d, tau, t, y, sigma2 = map(_f64, (a, b, c, d, tau, t, y, sigma2))
        mean_ = np.empty(len(tau))
        cov_ = np.empty((len(tau), len(tau))) if with_covariance else None
        info = ctypes.c_int32()
        _lib.check(_lib.lib().pioran_dense_predict(self._h, len(t), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(t), _ptr(y),
                                                   _ptr(sigma2), len(tau), _ptr(tau), _ptr(mean_), _ptr(cov_),
                                                   ctypes.byref(info)), self._h)
        res = (mean_, cov_) if with_covariance else mean_
        return (res, info.value) if return_info else res

    def dense_covariance(self, a, b, c, d, t, sigma2):
        a, b, c, d, t, sigma2 = map(_f64, (a, b, c, d, t, sigma2))
        K = np.empty((len(t), len(t)), dtype=np.float64)
        _lib.check(_lib.lib().pioran_dense_covariance(self._h, len(t), len(a), _ptr(a), _ptr(b), _ptr(c), _ptr(d),
                                                      _ptr(t), _ptr(sigma2), _ptr(K)), self._h)
        return K  # symmetric, so row/column-major does not matter


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Dataset:
    """A time series resident in HBM (pioran_ds): upload (t, y, sigma2) once, evaluate many batches."""

    def __init__(self, t, y, sigma2, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        t, y, sigma2 = map(_f64, (t, y, sigma2))
        if not (len(t) == len(y) == len(sigma2)) or len(t) < 1:
            raise ValueError("t, y, sigma2 must be non-empty and of equal length")
        self.N = len(t)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().pioran_dataset_create(self.ctx._h, self.N, _ptr(t), _ptr(y), _ptr(sigma2),
                                                    ctypes.byref(h)), self.ctx._h)
        self._h = h
        self.J = None

    def close(self):
        if getattr(self, "_h", None) and self.ctx._h:
            _lib.lib().pioran_dataset_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, c, d, real_term=None):
        """Declare the (c, d) shared by the following device-pointer batches; builds the table."""
        c, d = _f64(c), _f64(d)
        rt = None if real_term is None else np.ascontiguousarray(real_term, dtype=np.int32)
        _lib.check(_lib.lib().pioran_dataset_prepare(self._h, len(c), _ptr(c), _ptr(d), _ptr(rt)), self.ctx._h)
        self.J = len(c)

    def logl_batch(self, A, Bc, C, Dd, mu=None, nu=None, Y=None, S2=None, shift=None, return_status=False):
        """B log-likelihoods, host arrays.  A, Bc: (B, J); C, Dd: (J,) shared or (B, J) per draw.
        shift (B,): the data set holds raw flux and yerr**2; draw b is evaluated on log(y - shift_b) with
        variances sigma2 / (y - shift_b)**2, transformed on the device (docs/src/ultranest.md:199-205)."""
        A, Bc, C, Dd = map(_f64, (A, Bc, C, Dd))
        if A.ndim != 2 or A.shape != Bc.shape:
            raise ValueError("A and Bc must be (B, J) arrays of equal shape")
        B, J = A.shape
        cd_shared = C.ndim == 1
        if C.shape != Dd.shape or C.shape != ((J,) if cd_shared else (B, J)):
            raise ValueError("C, Dd must be (J,) or (B, J)")
        mu = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        if (Y is None) != (S2 is None):
            raise ValueError("Y and S2 must be given together")
        if shift is not None:
            if Y is not None:
                raise ValueError("give either shift or (Y, S2)")
            shift = _f64(np.broadcast_to(shift, (B,)))
            out = np.empty(B)
            st = np.zeros(B, dtype=np.int32)
            _lib.check(_lib.lib().pioran_celerite_logl_batch_shift(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd),
                                                                   int(cd_shared), _ptr(mu), _ptr(nu), _ptr(shift),
                                                                   _ptr(out), _ptr(st)), self.ctx._h)
            return (out, st) if return_status else out
        if Y is not None:
            Y, S2 = _f64(Y), _f64(S2)
            if Y.shape != (B, self.N) or S2.shape != (B, self.N):
                raise ValueError("Y, S2 must be (B, N)")
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        _lib.check(_lib.lib().pioran_celerite_logl_batch(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd),
                                                         int(cd_shared), _ptr(mu), _ptr(nu), _ptr(Y), _ptr(S2),
                                                         _ptr(out), _ptr(st)), self.ctx._h)
        return (out, st) if return_status else out

    def predict(self, A, Bc, C, Dd, tau, mu=None, nu=None, return_status=False):
        """Posterior mean at the times tau (M,) for B coefficient sets sharing (C, Dd): pred of
        src/celerite_solver.jl:363-483 (+ the constant mean mu_b).  Returns (B, M)."""
        A, Bc, C, Dd, tau = map(_f64, (A, Bc, C, Dd, tau))
        if A.ndim != 2 or A.shape != Bc.shape or C.shape not in ((A.shape[1],), A.shape) or Dd.shape != C.shape or tau.ndim != 1:
            raise ValueError("A, Bc must be (B, J), C, Dd (J,) or (B, J) and tau (M,)")
        B, J = A.shape
        mu = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        out = np.empty((B, len(tau)))
        st = np.zeros(B, dtype=np.int32)
        _lib.check(_lib.lib().pioran_celerite_predict(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(C.ndim == 1), _ptr(mu),
                                                      _ptr(nu), len(tau), _ptr(tau), _ptr(out), _ptr(st)), self.ctx._h)
        return (out, st) if return_status else out

    def logl_grad(self, A, Bc, C, Dd, mu=None, nu=None, series_grad=False, shift=None, cd_grad=True):
        """log L and its gradient for B draws: returns a dict with logl (B,), status, grad_a, grad_b, grad_c, grad_d (B, J),
        grad_mu, grad_nu (B,) and, with series_grad, grad_y, grad_sigma2 (B, N).  C, Dd: (J,) shared by the draws or
        (B, J) per draw (QPO / CARMA / free Celerite terms).  grad_c / grad_d are per draw also when (C, Dd) are shared.
        shift (B,): the shifted log-flux models (the data set holds raw flux and yerr**2); adds grad_shift (B,)."""
        A, Bc, C, Dd = map(_f64, (A, Bc, C, Dd))
        if A.ndim != 2 or A.shape != Bc.shape:
            raise ValueError("A, Bc must be (B, J)")
        B, J = A.shape
        cd_shared = C.ndim == 1
        if C.shape != Dd.shape or C.shape != ((J,) if cd_shared else (B, J)):
            raise ValueError("C, Dd must be (J,) or (B, J)")
        mu_ = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu_ = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        out = np.empty(B); st = np.zeros(B, dtype=np.int32)
        ga, gb = np.empty((B, J)), np.empty((B, J))
        gc = np.empty((B, J)) if cd_grad else None
        gd = np.empty((B, J)) if cd_grad else None
        gnu, gmu = np.empty(B), np.empty(B)
        L = _lib.lib()
        if shift is not None:
            shift = _f64(np.broadcast_to(shift, (B,)))
            gsh = np.empty(B)
            _lib.check(L.pioran_celerite_logl_grad_shift(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(cd_shared), _ptr(mu_),
                                                         _ptr(nu_), _ptr(shift), _ptr(out), _ptr(st), _ptr(ga), _ptr(gb), _ptr(gc),
                                                         _ptr(gd), _ptr(gnu), _ptr(gmu), _ptr(gsh)), self.ctx._h)
            return {"logl": out, "status": st, "grad_a": ga, "grad_b": gb, "grad_c": gc, "grad_d": gd, "grad_nu": gnu,
                    "grad_mu": gmu, "grad_shift": gsh, "grad_y": None, "grad_sigma2": None}
        gy = np.empty((B, self.N)) if series_grad else None
        gs = np.empty((B, self.N)) if series_grad else None
        _lib.check(L.pioran_celerite_logl_grad(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(cd_shared), _ptr(mu_), _ptr(nu_),
                                               _ptr(out), _ptr(st), _ptr(ga), _ptr(gb), _ptr(gc), _ptr(gd), _ptr(gnu), _ptr(gmu),
                                               _ptr(gy), _ptr(gs)), self.ctx._h)
        return {"logl": out, "status": st, "grad_a": ga, "grad_b": gb, "grad_c": gc, "grad_d": gd, "grad_nu": gnu, "grad_mu": gmu,
                "grad_y": gy, "grad_sigma2": gs}

    def logpdf_theta_grad(self, model, theta, norm, f_min, f_max, n_components=20, S_low=20.0, S_high=20.0, *,
                          is_integrated_power=True, basis_function="SHO", mu=None, nu=None, shift=None):
        """log L and its gradient with respect to the SAMPLED parameters of the reference's models
        (README.md:38-71: theta = PSD parameters, norm = variance, nu, mu): the device gradient w.r.t. (a, b, nu, mu)
        chained through `approx` on the host (approx_batch_vjp).  Returns a dict: logl, status, grad_theta (B, P),
        grad_norm, grad_nu, grad_mu (B,)."""
        from .psd import approx_batch, approx_batch_vjp
        theta = np.atleast_2d(_f64(theta))
        A, Bc, C, Dd = approx_batch(model, theta, f_min, f_max, n_components, norm, S_low, S_high,
                                    is_integrated_power=is_integrated_power, basis_function=basis_function)
        g = self.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu, shift=shift, cd_grad=False)   # (c, d) are fixed by the spectral grid: windowed reverse mode
        gth, gnorm = approx_batch_vjp(model, theta, f_min, f_max, n_components, norm, g["grad_a"], g["grad_b"], S_low, S_high,
                                      is_integrated_power=is_integrated_power, basis_function=basis_function)
        return {"logl": g["logl"], "status": g["status"], "grad_theta": gth, "grad_norm": gnorm,
                "grad_nu": g["grad_nu"] if nu is not None else None, "grad_mu": g["grad_mu"] if mu is not None else None,
                "grad_shift": g.get("grad_shift")}

    def logl_batch_dev(self, B, dA, dBc, dmu=0, dnu=0, dY=0, dS2=0, dout=0, dstatus=0):
        """Device-pointer (int addresses) asynchronous variant; (c, d) from prepare()."""
        v = ctypes.c_void_p
        _lib.check(_lib.lib().pioran_celerite_logl_batch_dev(self._h, int(B), v(dA), v(dBc), v(dmu or None),
                                                             v(dnu or None), v(dY or None), v(dS2 or None), v(dout),
                                                             v(dstatus or None)), self.ctx._h)


    def logpdf_theta(self, model, theta, norm, f_min, f_max, n_components=20, S_low=20.0, S_high=20.0, *,
                     is_integrated_power=True, basis_function="SHO", mu=None, nu=None, shift=None, qpo=None,
                     return_status=False, return_coefs=False):
        """theta -> log L in one call, `approx` running on the device (pioran_logpdf_batch_theta).
        model: SingleBendingPowerLaw / DoubleBendingPowerLaw (the class); theta: (B, 3 | 5); norm: scalar or (B,)
        — same meaning as the arguments of approx (src/psd.jl:214-289); mu, nu, shift as in logl_batch.
        qpo: (B, n_qpo, 3) or (B, 3) = (S0, f0, Q) of the QPO features added to the continuum (src/psd.jl:15-27, 228-241)."""
        from .psd import DoubleBendingPowerLaw, SingleBendingPowerLaw
        mid = {SingleBendingPowerLaw: 0, DoubleBendingPowerLaw: 1}.get(model)
        if mid is None:
            raise ValueError("model must be SingleBendingPowerLaw or DoubleBendingPowerLaw")
        if basis_function not in ("SHO", "DRWCelerite"):
            raise ValueError("Basis function" + basis_function + "not implemented")
        theta = _f64(np.atleast_2d(theta))
        B, P = theta.shape
        if P != (3 if mid == 0 else 5):
            raise ValueError("theta has the wrong number of columns for this model")
        norm = _f64(np.broadcast_to(norm, (B,)))
        mu = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        shift = None if shift is None else _f64(np.broadcast_to(shift, (B,)))
        basis = 0 if basis_function == "SHO" else 1
        n_qpo = 0
        if qpo is not None:
            qpo = _f64(np.asarray(qpo, dtype=np.float64).reshape(B, -1, 3))
            n_qpo = qpo.shape[1]
        Jt = n_components * (1 if basis == 0 else 2) + n_qpo
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        Ao = np.empty((B, Jt)) if return_coefs else None
        Bo = np.empty((B, Jt)) if return_coefs else None
        _lib.check(_lib.lib().pioran_logpdf_batch_theta(self._h, B, mid, int(n_components), basis, int(bool(is_integrated_power)),
                                                        float(f_min), float(f_max), float(S_low), float(S_high), _ptr(theta),
                                                        _ptr(norm), _ptr(mu), _ptr(nu), _ptr(shift), n_qpo, _ptr(qpo), _ptr(out), _ptr(st),
                                                        _ptr(Ao), _ptr(Bo)), self.ctx._h)
        res = (out, st) if return_status else (out,)
        if return_coefs:
            res = res + (Ao, Bo)
        return res if len(res) > 1 else res[0]

    def logl_batch_shift_dev(self, B, dA, dBc, dmu=0, dnu=0, dshift=0, dout=0, dstatus=0):
        """Device-pointer asynchronous variant of the shifted-log-flux batch; (c, d) from prepare()."""
        v = ctypes.c_void_p
        _lib.check(_lib.lib().pioran_celerite_logl_batch_shift_dev(self._h, int(B), v(dA), v(dBc), v(dmu or None),
                                                                   v(dnu or None), v(dshift), v(dout),
                                                                   v(dstatus or None)), self.ctx._h)


class Farm:
    """In-process farm over several GPUs (pioran_farm): one context + resident data set per listed device, one host
    thread per device per call, contiguous shards of the draws.  For launcher-less hosts (a single Julia / Python
    process driving a node); the process-per-GPU route is pioran.jl_amd/farm.py."""

    def __init__(self, devices, t, y, sigma2):
        t, y, sigma2 = map(_f64, (t, y, sigma2))
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().pioran_farm_create(len(dev), _ptr(dev), len(t), _ptr(t), _ptr(y), _ptr(sigma2), ctypes.byref(h)))
        self._h = h
        self.N = len(t)

    def __len__(self):
        return _lib.lib().pioran_farm_size(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().pioran_farm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def logl_batch(self, A, Bc, C, Dd, mu=None, nu=None, shift=None, Y=None, S2=None, return_status=False):
        """Y, S2: per-draw series (B, N) — e.g. y - mean_b(t) for a CustomMean model
        (examples/ultranest/single_pl_periodicity.jl:115); exclusive with shift."""
        A, Bc, C, Dd = map(_f64, (A, Bc, C, Dd))
        B, J = A.shape
        cd_shared = C.ndim == 1
        mu = None if mu is None else _f64(np.broadcast_to(mu, (B,)))
        nu = None if nu is None else _f64(np.broadcast_to(nu, (B,)))
        shift = None if shift is None else _f64(np.broadcast_to(shift, (B,)))
        out = np.empty(B)
        st = np.zeros(B, dtype=np.int32)
        if (Y is None) != (S2 is None) or (Y is not None and shift is not None):
            raise ValueError("Y and S2 come together, and not with shift")
        if Y is not None:
            Y, S2 = _f64(Y), _f64(S2)
            if Y.shape != (B, self.N) or S2.shape != (B, self.N):
                raise ValueError("Y, S2 must be (B, N)")
            _lib.check(_lib.lib().pioran_farm_logl_batch_series(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(cd_shared),
                                                                _ptr(mu), _ptr(nu), _ptr(Y), _ptr(S2), _ptr(out), _ptr(st)))
        else:
            _lib.check(_lib.lib().pioran_farm_logl_batch(self._h, B, J, _ptr(A), _ptr(Bc), _ptr(C), _ptr(Dd), int(cd_shared),
                                                         _ptr(mu), _ptr(nu), _ptr(shift), _ptr(out), _ptr(st)))
        return (out, st) if return_status else out


# ---------------------------------------------------------------------------------------------
# mean functions (AbstractGPs ConstMean / ZeroMean / CustomMean as used at scalable_GP.jl:164)
# ---------------------------------------------------------------------------------------------
class CustomMean:
    def __init__(self, f):
        self.f = f

    def __call__(self, x):
        return np.asarray(self.f(np.asarray(x, float)), float)


def _mean_vector(mean, x):
    if isinstance(mean, CustomMean):
        return mean(x)
    return np.full(len(x), float(mean))


class ScalableGP:
    """ScalableGP(kernel) | ScalableGP(mu, kernel) | ScalableGP(mu, kernel, solver)   scalable_GP.jl:24-40."""

    def __init__(self, *args):
        if len(args) == 1:
            mean, kernel, solver = 0.0, args[0], "celerite"
        elif len(args) == 2:
            mean, kernel, solver = args[0], args[1], "celerite"
        elif len(args) == 3:
            mean, kernel, solver = args
        else:
            raise TypeError("ScalableGP(kernel) | ScalableGP(mean, kernel[, solver])")
        if not isinstance(kernel, SemiSeparable):
            raise TypeError("kernel must be a SemiSeparable covariance")
        self.mean, self.kernel, self.solver = mean, kernel, str(solver).lstrip(":")

    def __call__(self, t, sigma2=None):
        t = _f64(t)
        s2 = np.zeros(len(t)) if sigma2 is None else _f64(np.broadcast_to(sigma2, t.shape))
        return FiniteScalableGP(self, t, s2)


class FiniteScalableGP:
    """f(t, sigma2): AbstractGPs.FiniteGP{<:ScalableGP} (scalable_GP.jl:42)."""

    def __init__(self, f: ScalableGP, x, sigma2):
        self.f, self.x, self.sigma2 = f, x, sigma2


class PosteriorGP:
    """posterior(f(t, sigma2), y)   src/scalable_GP.jl:44-55."""

    def __init__(self, fx: "FiniteScalableGP", y):
        self.f, self.y = fx, _f64(y).reshape(-1)
        if len(self.y) != len(fx.x):
            raise ValueError("y must have one value per time stamp")


def posterior(fx: "FiniteScalableGP", y) -> PosteriorGP:
    return PosteriorGP(fx, y)


def predict(cov: SemiSeparable, tau, t, y, sigma2, ctx: Context | None = None):
    """predict(cov, tau, t, y, sigma2): posterior mean of the zero-mean GP   src/celerite_solver.jl:348-361."""
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in cov.celerite_coefs())
    ds = Dataset(t, y, sigma2, ctx)
    try:
        return ds.predict(a[None, :], b[None, :], c, d, tau)[0]
    finally:
        ds.close()


def mean(fp: PosteriorGP, tau=None, ctx: Context | None = None):
    """mean(fp[, tau])   src/scalable_GP.jl:64-72, 90-91: predict on y - mean(t), plus mean(tau)."""
    x = fp.f.x
    tau = x if tau is None else _f64(tau).reshape(-1)
    y0 = fp.y - _mean_vector(fp.f.f.mean, x)
    return predict(fp.f.f.kernel, tau, x, y0, fp.f.sigma2, ctx=ctx) + _mean_vector(fp.f.f.mean, tau)


def predict_cov(cov_fn: SemiSeparable, tau, t, sigma2, ctx: Context | None = None):
    """predict_cov(cov, tau, t, sigma2)   src/direct_solver.jl:28-69 (dense, MFMA Cholesky of the augmented matrix)."""
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in cov_fn.celerite_coefs())
    K, info = (ctx or default_context()).dense_predict_cov(a, b, c, d, tau, t, sigma2, return_info=True)
    if info != 0:
        raise np.linalg.LinAlgError(f"matrix is not positive definite; Cholesky factorization failed at pivot {info}")
    return K


def predict_direct(cov_fn: SemiSeparable, tau, t, y, sigma2, with_covariance=False, ctx: Context | None = None):
    """predict_direct(cov, tau, t, y, sigma2[, with_covariance])   src/direct_solver.jl:75-119."""
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in cov_fn.celerite_coefs())
    res, info = (ctx or default_context()).dense_predict(a, b, c, d, tau, t, y, sigma2, with_covariance, return_info=True)
    if info != 0:
        raise np.linalg.LinAlgError(f"matrix is not positive definite; Cholesky factorization failed at pivot {info}")
    return res


def cov(fp: PosteriorGP, tau=None, ctx: Context | None = None):
    """cov(fp[, tau])   src/scalable_GP.jl:73-78, 95-96."""
    tau = fp.f.x if tau is None else _f64(tau).reshape(-1)
    return predict_cov(fp.f.f.kernel, tau, fp.f.x, fp.f.sigma2, ctx=ctx)


def std(fp: PosteriorGP, tau=None, ctx: Context | None = None):
    """std(fp[, tau]) = sqrt.(diag(cov))   src/scalable_GP.jl:103-104."""
    return np.sqrt(np.diag(cov(fp, tau, ctx=ctx)))


def rand_posterior(rng, fp: PosteriorGP, tau=None, n: int = 1, ctx: Context | None = None):
    """rand(rng, fp[, tau], N): draws from MvNormal(mean, cov)   src/scalable_GP.jl:106-129.  Returns (len(tau), n)."""
    tau = fp.f.x if tau is None else _f64(tau).reshape(-1)
    m, K = mean(fp, tau, ctx=ctx), cov(fp, tau, ctx=ctx)
    L = np.linalg.cholesky(K + 1e-14 * np.trace(K) / len(tau) * np.eye(len(tau)))
    return m[:, None] + L @ rng.standard_normal((len(tau), n))


def simulate(rng, cov: SemiSeparable, tau, sigma2, ctx: Context | None = None):
    """simulate(rng, cov, tau, sigma2)   src/celerite_solver.jl:497-513: one realisation; `rng` is a numpy Generator
    supplying the standard normals (the reference's randn(rng, N), :528)."""
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in cov.celerite_coefs())
    tau = _f64(tau).reshape(-1)
    q = rng.standard_normal(len(tau))
    s2 = _f64(np.broadcast_to(sigma2, tau.shape))
    return (ctx or default_context()).simulate(a[None, :], b[None, :], c, d, tau, s2, q[None, :])[0]


def rand(rng, fx: "FiniteScalableGP", t=None, ctx: Context | None = None):
    """rand(rng, f(t, sigma2)) / rand(rng, f(t, sigma2), t')   src/scalable_GP.jl:137-158 (zero variance at new times)."""
    if t is None:
        return simulate(rng, fx.f.kernel, fx.x, fx.sigma2, ctx=ctx) + _mean_vector(fx.f.mean, fx.x)
    t = _f64(t).reshape(-1)
    return simulate(rng, fx.f.kernel, t, np.zeros(len(t)), ctx=ctx) + _mean_vector(fx.f.mean, t)


def logpdf(fx: FiniteScalableGP, Y, ctx: Context | None = None):
    """Distributions.logpdf(f::FiniteScalableGP, Y)   src/scalable_GP.jl:162-166."""
    Y = _f64(Y).reshape(-1)
    y = Y - _mean_vector(fx.f.mean, fx.x)
    return log_likelihood(fx.f.kernel, fx.x, y, fx.sigma2, solver=fx.f.solver, ctx=ctx)


def log_likelihood(cov: SemiSeparable, tau, y, sigma2, solver="celerite", ctx: Context | None = None):
    """log_likelihood(cov, tau, y, sigma2; solver)   src/celerite_solver.jl:262-294.
    Both solver symbols the reference accepts run the same HIP kernel (the reference's
    :celerite_matrix variant is experimental and untested, src/celerite_solver.jl:160-248)."""
    solver = str(solver).lstrip(":")
    if solver not in _SOLVERS:
        raise ValueError(f"solver {solver} not recognised, use either :celerite or :celerite_matrix")
    a, b, c, d = cov.celerite_coefs()
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in (a, b, c, d))
    return logl(a, b, c, d, tau, y, sigma2, ctx=ctx)


def logl(a, b, c, d, tau, y, sigma2, ctx: Context | None = None):
    """logl(a, b, c, d, tau, y, sigma2)   src/celerite_solver.jl:312-334."""
    val, st = (ctx or default_context()).logl(a, b, c, d, tau, y, sigma2, return_status=True)
    if st == 2 and not np.isfinite(val):
        # the reference throws DomainError from log(D[1] < 0) (celerite_solver.jl:126)
        raise ValueError("log-likelihood is not finite: the covariance matrix is not positive definite")
    return val


def log_likelihood_direct(cov: SemiSeparable, t, y, sigma2, ctx: Context | None = None):
    """log_likelihood_direct(cov, t, y, sigma2): dense solver, returns the POSITIVE NLL like the
    reference (src/direct_solver.jl:19).  Raises like PosDefException when K is not PD."""
    a, b, c, d = (np.real(np.atleast_1d(v)) for v in cov.celerite_coefs())
    val, info = (ctx or default_context()).dense_nll(a, b, c, d, t, y, sigma2, return_info=True)
    if info != 0:
        raise np.linalg.LinAlgError(f"matrix is not positive definite; Cholesky factorization failed at pivot {info}")
    return val


def logpdf_batch(ds: Dataset, A, Bc, C, Dd, mu=None, nu=None, Y=None, S2=None, shift=None, return_status=False):
    """B independent logpdf evaluations on one data set (nested-sampling live points / MCMC walkers)."""
    return ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, Y=Y, S2=S2, shift=shift, return_status=return_status)
