"""ctypes binding of libpioran_hip.so (the C ABI of include/pioran_hip.h).

The library is the product: if it is missing or fails to load, importing this module's `lib()` raises —
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# PIORAN_HIP_LIB: another build of the same library (A/B timing of two kernel versions on one GPU box, tools/ab_bench.py)
LIB_PATH = Path(os.environ["PIORAN_HIP_LIB"]) if os.environ.get("PIORAN_HIP_LIB") else _PKG / "libpioran_hip.so"

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_void_p = ctypes.c_void_p
i64 = ctypes.c_int64

# name -> (restype, argtypes); every symbol declared in include/pioran_hip.h
SIGNATURES = {
    "pioran_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "pioran_last_hip_error": (ctypes.c_char_p, [c_void_p]),
    "pioran_abi_version": (ctypes.c_int, []),
    "pioran_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_void_p)]),
    "pioran_ctx_create_on_stream": (ctypes.c_int, [ctypes.c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "pioran_ctx_destroy": (ctypes.c_int, [c_void_p]),
    "pioran_ctx_synchronize": (ctypes.c_int, [c_void_p]),
    "pioran_ctx_trim": (ctypes.c_int, [c_void_p]),
    "pioran_ctx_set_option": (ctypes.c_int, [c_void_p, ctypes.c_char_p, ctypes.c_char_p]),
    "pioran_ctx_event_record": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "pioran_ctx_event_elapsed_ms": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]),
    "pioran_dataset_create": (ctypes.c_int, [c_void_p, i64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]),
    "pioran_dataset_destroy": (ctypes.c_int, [c_void_p]),
    "pioran_dataset_prepare": (ctypes.c_int, [c_void_p, i64, c_void_p, c_void_p, c_void_p]),
    "pioran_celerite_logl": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 7 + [c_void_p, c_void_p]),
    "pioran_celerite_logl_batch": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p]),
    "pioran_celerite_logl_batch_dev": (ctypes.c_int, [c_void_p, i64] + [c_void_p] * 8),
    "pioran_celerite_logl_batch_dev_cd": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 10),
    "pioran_celerite_logl_batch_shift": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                                        ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pioran_celerite_logl_batch_shift_dev": (ctypes.c_int, [c_void_p, i64] + [c_void_p] * 7),
    "pioran_logpdf_batch_theta": (ctypes.c_int, [c_void_p, i64, ctypes.c_int, i64, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, i64, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p]),
    "pioran_celerite_predict": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int, c_void_p,
                                               c_void_p, i64, c_void_p, c_void_p, c_void_p]),
    "pioran_celerite_logl_grad": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 4 + [ctypes.c_int] + [c_void_p] * 12),
    "pioran_celerite_logl_grad_shift": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 4 + [ctypes.c_int] + [c_void_p] * 12),
    "pioran_celerite_simulate": (ctypes.c_int, [c_void_p, i64, i64, i64] + [c_void_p] * 4 + [ctypes.c_int] + [c_void_p] * 4),
    "pioran_celerite_config_name": (ctypes.c_char_p, [i64]),
    "pioran_ctx_fp64_probe": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]),
    "pioran_tile_choice": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "pioran_farm_create": (ctypes.c_int, [ctypes.c_int, c_void_p, i64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]),
    "pioran_farm_destroy": (ctypes.c_int, [c_void_p]),
    "pioran_farm_size": (ctypes.c_int, [c_void_p]),
    "pioran_farm_logl_batch": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pioran_farm_logl_batch_series": (ctypes.c_int, [c_void_p, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int,
                                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pioran_dense_nll": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 7 + [c_void_p, c_void_p]),
    "pioran_dense_nll_batch": (ctypes.c_int, [c_void_p, i64, i64, i64, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pioran_dense_nll_timed": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 7 + [c_void_p, c_void_p, c_void_p]),
    "pioran_dense_predict_cov": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 6 + [i64, c_void_p, c_void_p, c_void_p]),
    "pioran_dense_predict": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 7 + [i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pioran_dense_covariance": (ctypes.c_int, [c_void_p, i64, i64] + [c_void_p] * 6 + [c_void_p]),
}

_lib = None


class PioranHipError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's): whichever
    copy is loaded first serves every later user, and torch finds no GPU when it ends up on the other one.  If torch is
    installed but not imported yet, load ITS copy now, so that `import torch` after the first pioran call still works
    (bench.py, farm.py and the tests use torch for device buffers and torch.distributed).  Without torch: nothing to do."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        cand = Path(d) / "lib" / "libamdhip64.so"
        if cand.exists():
            try:
                ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def lib():
    """Loads libpioran_hip.so; raises if it has not been built (python pioran.jl_amd/build.py)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise PioranHipError(
                f"{LIB_PATH} not found: build the HIP library first (python pioran.jl_amd/build.py or "
                f"__graft_entry__.build()). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, ctx=None):
    if rc != 0:
        msg = lib().pioran_strerror(rc).decode()
        if ctx is not None:
            detail = lib().pioran_last_hip_error(ctx).decode()
            if detail:
                msg += f" ({detail})"
        raise PioranHipError(f"pioran_hip error {rc}: {msg}")
