"""CPU tests (-m "not gpu"): the C-ABI library builds for gfx950, loads, and exports every symbol that
include/pioran_hip.h declares; entry points reject bad arguments without a GPU; the product fails
loudly (no CPU fallback) when no GPU is present."""
import ctypes
import re
from pathlib import Path

import pytest

import pioran_jl_amd as pj

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "pioran_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pioran_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    names = _declared_symbols()
    assert len(names) >= 26
    L = ctypes.CDLL(str(pj._lib.LIB_PATH))
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pioran_hip.h but not exported"
    assert sorted(pj._lib.SIGNATURES) == names          # the ctypes binding covers exactly the header


def test_library_is_gfx950_code_object():
    data = pj._lib.LIB_PATH.read_bytes()
    assert b"gfx950" in data and b"celerite_scan_kernel" in data


def test_version_and_strerror():
    L = pj._lib.lib()
    assert L.pioran_abi_version() == 7
    assert L.pioran_strerror(0) == b"ok"
    assert L.pioran_strerror(-4) == b"unsupported size"
    assert L.pioran_celerite_config_name(40) == b"rpl3_cbr2_nsrc7_p"   # column-paired variant for the standard row map
    assert L.pioran_celerite_config_name(60).startswith(b"rpl4_cbr4")
    assert L.pioran_celerite_config_name(128) == b"wide" and L.pioran_celerite_config_name(80) == b"wide"   # lean latency kernel, 80 .. 143 rows
    assert L.pioran_celerite_config_name(144) == b"fallback"


def test_argument_validation_without_gpu():
    L = pj._lib.lib()
    assert L.pioran_ctx_create(0, None) == -1
    assert L.pioran_ctx_destroy(None) == -1
    assert L.pioran_dataset_create(None, 10, None, None, None, None) == -1
    assert L.pioran_celerite_logl_batch(None, 1, 1, None, None, None, None, 1, None, None, None, None, None, None) == -1
    assert L.pioran_celerite_logl_batch_dev(None, 1, None, None, None, None, None, None, None, None) == -1
    assert L.pioran_ctx_set_option(None, b"no_wide", b"1") == -1
    assert L.pioran_dense_nll_timed(None, 4, 1, None, None, None, None, None, None, None, None, None, None) == -1


def test_launch_path_never_reads_the_environment():
    """Diagnostic switches are context state (read from the environment once, at context creation)."""
    csrc = ROOT / "pioran.jl_amd" / "csrc"
    hits = [(f.name, i + 1) for f in csrc.glob("*") for i, line in enumerate(f.read_text().splitlines()) if "getenv(" in line]
    assert hits and all(name == "capi.hip" for name, _ in hits)
    capi = (csrc / "capi.hip").read_text()
    body = capi[capi.index("static int ctx_create_impl"):capi.index("int pioran_ctx_create(int device")]
    assert capi.count("getenv(") == body.count("getenv(")     # every read sits inside context creation


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pj._lib.PioranHipError):
        pj.Context(0)
    with pytest.raises(pj._lib.PioranHipError):
        pj.logl([1.0], [0.0], [0.5], [0.0], [0.0, 1.0], [0.1, 0.2], [0.01, 0.01])


def test_product_never_imports_oracle():
    pkg = ROOT / "pioran.jl_amd"
    for f in list(pkg.glob("*.py")) + list((pkg / "csrc").glob("*")):
        txt = f.read_text(errors="ignore")
        assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_shipped_library_is_what_build_py_makes(tmp_path):
    """The library in the tree is newer than every source it is built from, its objects are the ones build.py links, and a fresh compile
    of one small source with build.py's own flags reproduces the shipped object's size (round 4 shipped a library linked from an object
    that a diagnostic tool had overwritten: same source and flags, but nothing checked it)."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("_pioran_build_chk", ROOT / "pioran.jl_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib_m = b.LIB.stat().st_mtime
    deps = [b.CSRC / s for s in b.SOURCES] + list(b.HEADERS)
    stale = [d.name for d in deps if d.stat().st_mtime > lib_m]
    assert not stale, f"libpioran_hip.so is older than {stale}: run python pioran.jl_amd/build.py"
    for s in b.SOURCES:
        obj = b.OBJ / (Path(s).stem + ".o")
        assert obj.exists() and obj.stat().st_mtime <= lib_m + 1.0, obj
        assert obj.stat().st_mtime >= (b.CSRC / s).stat().st_mtime, f"{obj.name} is older than its source"
    assert "celerite_tile.hip" in b.SOURCES and b"celerite_tile_kernel" in b.LIB.read_bytes()
    small = "celerite_fallback.hip"
    fresh = tmp_path / "fresh.o"
    subprocess.run([b.hipcc(), *b.FLAGS, "-c", str(b.CSRC / small), "-o", str(fresh)], check=True)
    assert fresh.stat().st_size == (b.OBJ / "celerite_fallback.o").stat().st_size
    # no experiment code in the product: the persistent dense chain and the two-wavefront scan shape need -DPIORAN_EXPERIMENTS
    data = b.LIB.read_bytes()
    assert b"dense_crit_chain_kernel" not in data and b"rpl5_cbr4_nsrc2_w2" not in data


def test_probe_and_option_arguments_without_a_gpu():
    """Argument validation of the ABI-7 additions happens before any GPU call: NULL context / NULL output / out-of-range arguments are
    PIORAN_ERR_ARG (pioran_ctx_fp64_probe, pioran_ctx_set_option)."""
    import ctypes
    import pioran_jl_amd as pj
    L = pj._lib.lib()
    out = ctypes.c_double(0.0)
    assert L.pioran_ctx_fp64_probe(None, 2, 10.0, ctypes.byref(out)) != 0
    assert L.pioran_ctx_set_option(None, b"no_tp", b"1") != 0
