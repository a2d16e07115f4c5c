"""CPU tests (-m "not gpu") of the host-side mirror of the reference interface (pioran.jl_amd/*.py):
kernel containers, `+`/scaling algebra, approx / approx_batch — against the reference's literals and the
pinned oracle.  Nothing here touches the GPU."""
import json
import math

import numpy as np
import pytest

import pioran_jl_amd as pj
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
from oracle import oracle as O


@pytest.fixture(scope="module")
def lit(golden_dir):
    return json.loads((golden_dir / "reference_literals.json").read_text())


def test_term_kernels_closed_forms():
    """test/test_covariancefunctions.jl:3-30."""
    t = np.linspace(0, 10, 500)
    e = pj.Exp(1.0, 2.4)
    np.testing.assert_allclose(e(t, 0.0), np.exp(-t * 2.4) / 2, rtol=1e-15)
    a, b, c, d = 1.3, 4.0, 0.5, 3.2
    t = np.linspace(0, 25, 500)
    np.testing.assert_allclose(pj.Celerite(a, b, c, d)(t, 0.0), np.exp(-c * t) * (a * np.cos(d * t) + b * np.sin(d * t)),
                               rtol=1e-14, atol=1e-15)
    A, w0, Q = 1.5, 2 * math.pi * 0.23, 1 / math.sqrt(2)
    t = np.linspace(0, 15, 500)
    eta = math.sqrt(abs(1 - 1 / (4 * Q ** 2)))
    np.testing.assert_allclose(pj.SHO(A, w0, Q)(t, 0.0),
                               A * np.exp(-w0 * t / 2 / Q) * (np.cos(eta * w0 * t) + np.sin(eta * w0 * t) / (2 * eta * Q)),
                               rtol=1e-14, atol=1e-15)


def test_celerite_coefs_literals(lit):
    assert pj.celerite_coefs(pj.Celerite(*lit["coefs_celerite"]["args"])) == lit["coefs_celerite"]["expected"]
    assert pj.celerite_coefs(pj.Exp(*lit["coefs_exp"]["args"])) == lit["coefs_exp"]["expected"]
    s = lit["coefs_sho"]
    w0 = 2 * math.pi * s["w0_over_2pi"]
    assert pj.celerite_coefs(pj.SHO(s["A"], w0, 1 / math.sqrt(2))) == [s["A"], s["A"], math.sqrt(2) / 2 * w0,
                                                                        math.sqrt(2) / 2 * w0]
    with pytest.raises(ValueError, match="SHO with Q≠1/√2 not implemented yet"):
        pj.celerite_coefs(pj.SHO(s["A"], w0, 0.5))


def _mk(term):
    return {"Exp": pj.Exp, "Celerite": pj.Celerite}[term[0]](*term[1:])


def test_sum_and_scale_algebra(lit):
    """test/test_acvf.jl:3-33: `+` concatenation order and scaling of the amplitudes only."""
    g = lit["acvf_sum_scaled_exp"]
    e = g["scale"] * (_mk(g["terms"][0]) + _mk(g["terms"][1]))
    got = [list(v) for v in pj.celerite_coefs(e)]
    assert got == g["expected"]
    g = lit["acvf_large_sum"]
    k = _mk(g["terms"][0]) + _mk(g["terms"][1]) + _mk(g["terms"][2]) + _mk(g["terms"][3])
    assert [list(v) for v in pj.celerite_coefs(k)] == g["expected"]
    assert isinstance(k, pj.SumOfSemiSeparable) and isinstance(k, pj.SumOfTerms)
    t = np.linspace(0, 10, 500)
    e1, e2 = pj.Exp(1.0, 0.34), pj.Exp(2.4, 0.21)
    np.testing.assert_allclose((e1 + e2)(t, 0.0), e1(t, 0.0) + e2(t, 0.0), rtol=1e-15)
    np.testing.assert_allclose((12.5 * (e1 + e2))(t, 0.0), 12.5 * (e1(t, 0.0) + e2(t, 0.0)), rtol=1e-15)


def test_sum_of_celerite_container():
    """test/test_acvf.jl:35-64."""
    rng = np.random.default_rng(1234)
    a, b, c, d = 2 * rng.random(10), rng.random(10), rng.random(10), rng.random(10)
    C = pj.SumOfCelerite(a, b, c, d)
    assert isinstance(C, pj.SumOfTerms)
    for got, exp in zip(pj.celerite_coefs(C), (a, b, c, d)):
        assert (got == exp).all()
    assert C.cov == [pj.Celerite(*x) for x in zip(a, b, c, d)]
    S = pj.ScaledKernel(C, 3.0)
    assert (S.a == 3.0 * a).all() and (S.b == 3.0 * b).all() and (S.c == c).all() and (S.d == d).all()
    tau = np.linspace(0, 5, 40)
    np.testing.assert_allclose(C.kappa(tau), [O.kappa(a, b, c, d, x) for x in tau], rtol=1e-13)


def test_psd_models_and_amplitude_golden(lit):
    f = 10 ** np.linspace(-3, 2, 1000)
    P = pj.SingleBendingPowerLaw(0.3, 0.02, 2.93)
    assert (P(f) == (f / 0.02) ** (-0.3) / (1 + (f / 0.02) ** (2.93 - 0.3))).all()      # test/test_psd.jl:3-7
    Dm = pj.DoubleBendingPowerLaw(0.3, 0.02, 1.4, 10.2, 2.93)
    assert (Dm(f) == (f / 0.02) ** (-0.3) / (1 + (f / 0.02) ** (1.4 - 0.3)) / (1 + (f / 10.2) ** (2.93 - 1.4))).all()
    g = lit["psd_amplitudes"]
    sp, _ = pj.build_approx(g["J"], g["f0"], g["fM"])
    np.testing.assert_allclose(sp, g["f0"] * (g["fM"] / g["f0"]) ** (np.arange(g["J"]) / (g["J"] - 1)), rtol=1e-15)
    amp = pj.get_approx_coefficients(pj.SingleBendingPowerLaw(*g["params"]), g["f0"], g["fM"], n_components=g["J"])
    np.testing.assert_allclose(amp, g["expected"], rtol=1e-9)                              # test/test_psd.jl:32-39


@pytest.mark.parametrize("basis", ["SHO", "DRWCelerite"])
@pytest.mark.parametrize("integ", [True, False])
def test_approx_matches_oracle(basis, integ):
    for a1, f1, a2, var in [(0.2, 1.3e-2, 3.2, 1.32), (0.92, 0.5, 3.8, 0.21), (-0.2, 3.0, 1.6, 5.0)]:
        R = pj.approx(pj.SingleBendingPowerLaw(a1, f1, a2), 2e-3, 3.52e2, 25, var, is_integrated_power=integ,
                      basis_function=basis)
        ref = O.approx(lambda f: O.single_bending_power_law(f, a1, f1, a2), 2e-3, 3.52e2, 25, var,
                       is_integrated_power=integ, basis_function=basis)
        assert isinstance(R, pj.SumOfCelerite)
        for got, exp in zip(pj.celerite_coefs(R), ref):   # cond(spectral matrix) ~ 1e3-1e4 amplifies ulp-level
            np.testing.assert_allclose(got, exp, rtol=1e-9, atol=1e-14 * np.abs(exp).max())  # differences in the grid
        if not integ:
            assert R.a.sum() == pytest.approx(var, rel=1e-12)                              # test/test_psd.jl:100-153
        assert len(R.a) == (25 if basis == "SHO" else 50)


def test_approx_with_qpo_features():
    """test/test_psd.jl:206-285: one extra celerite term per QPO, continuum required."""
    PS = pj.SingleBendingPowerLaw(0.2, 1.3e-2, 3.2) + pj.QPO(2.0, 1e-2, 14.2) + pj.QPO(4.0, 1e-1, 4.2)
    R = pj.approx(PS, 2e-3, 3.52e2, 25, 1.32, is_integrated_power=False)
    ref = O.approx(lambda f: O.single_bending_power_law(f, 0.2, 1.3e-2, 3.2), 2e-3, 3.52e2, 25, 1.32,
                   is_integrated_power=False, qpo_features=[(2.0, 1e-2, 14.2), (4.0, 1e-1, 4.2)])
    assert len(R.a) == 27
    for got, exp in zip(pj.celerite_coefs(R), ref):
        np.testing.assert_allclose(got, exp, rtol=1e-9)
    R = pj.approx(pj.SingleBendingPowerLaw(0.2, 1.3e-2, 4.2) + pj.QPO(1.4, 1e-2, 10.2), 2e-3, 3.52e2, 25, 1.32,
                  is_integrated_power=True, basis_function="DRWCelerite")
    assert len(R.a) == 51
    with pytest.raises(AssertionError, match="ContinuumPowerSpectrum"):
        pj.approx(pj.QPO(1.4, 1e-2, 10.2), 2e-3, 3.52e2, 25, 1.0)
    with pytest.raises(ValueError, match="not implemented"):
        pj.approx(pj.SingleBendingPowerLaw(0.2, 1.3e-2, 4.2), 2e-3, 3.52e2, 25, 1.0, basis_function="foo")


@pytest.mark.parametrize("basis", ["SHO", "DRWCelerite"])
def test_approx_batch_equals_scalar(basis):
    rng = np.random.default_rng(3)
    B = 70
    th = np.column_stack([rng.uniform(-0.25, 2, B), np.exp(rng.uniform(np.log(1e-3), np.log(5), B)), rng.uniform(1.5, 4, B)])
    var = np.exp(rng.standard_normal(B))
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th, 1e-3, 5.0, 20, var, basis_function=basis)
    assert A.flags["C_CONTIGUOUS"] and Bc.flags["C_CONTIGUOUS"] and A.shape == (B, 20 if basis == "SHO" else 40)
    for i in (0, 13, 69):
        R = pj.approx(pj.SingleBendingPowerLaw(*th[i]), 1e-3, 5.0, 20, var[i], basis_function=basis)
        np.testing.assert_allclose(A[i], R.a, rtol=1e-9, atol=1e-14 * np.abs(R.a).max())
        np.testing.assert_allclose(Bc[i], R.b, rtol=1e-9, atol=1e-14 * np.abs(R.a).max())
        np.testing.assert_allclose(C, R.c, rtol=1e-15)
        np.testing.assert_allclose(Dd, R.d, rtol=1e-15)
    A2, _, _, _ = pj.approx_batch(pj.SingleBendingPowerLaw, th[:5], 1e-3, 5.0, 20, var[:5], basis_function=basis)
    np.testing.assert_allclose(A2, A[:5], rtol=1e-9, atol=1e-14)   # small-batch (loop) and vectorised PSD evaluation agree


def test_scalable_gp_construction():
    """test/test_scalablegp.jl:85-107."""
    R = pj.approx(pj.SingleBendingPowerLaw(0.2, 0.02, 3.1), 1e-4, 1e1, 30, 2.31, basis_function="SHO")
    f = pj.ScalableGP(R)
    fm = pj.ScalableGP(1.2, R)
    assert f.mean == 0.0 and fm.mean == 1.2 and f.solver == "celerite"
    assert isinstance(f.kernel, pj.SumOfCelerite)
    fx = fm(np.array([0.0, 1.0, 2.5]), np.array([0.1, 0.2, 0.1]))
    assert isinstance(fx, pj.FiniteScalableGP) and fx.sigma2.shape == (3,)
    with pytest.raises(TypeError):
        pj.ScalableGP(1.0, "not a kernel")


def test_approx_batch_vjp_matches_finite_differences():
    """Chain rule through approx (complex step on the host side of the gradient path)."""
    import pioran_jl_amd as pj
    rng = np.random.default_rng(0)
    th = np.column_stack([rng.uniform(0, 1.5, 5), 10 ** rng.uniform(-3, 0, 5), rng.uniform(2, 4, 5)])
    norm = rng.uniform(0.5, 2, 5)
    for basis in ("SHO", "DRWCelerite"):
        A, Bc, _, _ = pj.approx_batch(pj.SingleBendingPowerLaw, th, 1e-3, 5.0, 20, norm, basis_function=basis)
        ga, gb = rng.standard_normal(A.shape), rng.standard_normal(A.shape)
        gth, gn = pj.approx_batch_vjp(pj.SingleBendingPowerLaw, th, 1e-3, 5.0, 20, norm, ga, gb, basis_function=basis)
        h = 1e-6
        for k in range(3):
            e = np.zeros_like(th); e[:, k] = h * np.maximum(1, np.abs(th[:, k]))
            Ap, Bp, _, _ = pj.approx_batch(pj.SingleBendingPowerLaw, th + e, 1e-3, 5.0, 20, norm, basis_function=basis)
            Am, Bm, _, _ = pj.approx_batch(pj.SingleBendingPowerLaw, th - e, 1e-3, 5.0, 20, norm, basis_function=basis)
            fd = (ga * (Ap - Am) + gb * (Bp - Bm)).sum(1) / (2 * e[:, k])
            assert np.max(np.abs(fd - gth[:, k]) / (1 + np.abs(fd))) < 1e-7
        Ap, Bp, _, _ = pj.approx_batch(pj.SingleBendingPowerLaw, th, 1e-3, 5.0, 20, norm * (1 + h), basis_function=basis)
        fdn = (ga * (Ap - A) + gb * (Bp - Bc)).sum(1) / (norm * h)
        assert np.max(np.abs(fdn - gn) / (1 + np.abs(gn))) < 1e-6


def test_carma_kernel_coefficients(golden_dir):
    """CARMA(3, 2) -> (a, b, c, d): the reference's literal (test/test_carma.jl:55-69), the oracle, argument checks."""
    from oracle import oracle as O
    g = json.loads((golden_dir / "reference_literals.json").read_text())["carma32"]
    r = np.array([complex(*z) for z in g["r_alpha"]])
    k = pj.CARMA(g["p"], g["q"], r, g["beta"], g["norm"])
    got = k.celerite_coefs()
    for x, e in zip(got, g["expected"]):
        np.testing.assert_allclose(x, e, rtol=1e-12, atol=1e-14)
    for x, e in zip(got, O.carma_celerite_coefs(g["p"], r, g["beta"], g["norm"])):
        np.testing.assert_allclose(x, e, rtol=1e-13, atol=1e-15)
    # even order, variance normalisation, scaling, kappa(0) = sum(a)
    r4 = np.array([-0.1 + 0.7j, -0.1 - 0.7j, -0.4 + 0.2j, -0.4 - 0.2j])
    k4 = pj.CARMA(4, 1, r4, [1.0, 0.3], 2.0, False)
    a, b, c, d = k4.celerite_coefs()
    assert len(a) == 2 and abs(k4.kappa(0.0) - a.sum()) < 1e-14
    for x, e in zip(k4.celerite_coefs(), O.carma_celerite_coefs(4, r4, [1.0, 0.3], 2.0, False)):
        np.testing.assert_allclose(x, e, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose((3.0 * k4).celerite_coefs()[0], 3.0 * a, rtol=1e-14)
    assert isinstance(k.celerite_repr(), pj.SumOfCelerite)
    for bad in (dict(p=0, q=0, r=[], beta=[1.0]), dict(p=2, q=3, r=r4[:2], beta=[1, 2, 3, 4]),
                dict(p=3, q=1, r=r4[:2], beta=[1, 2]), dict(p=2, q=1, r=r4[:2], beta=[1.0])):
        with pytest.raises(ValueError):
            pj.CARMA(bad["p"], bad["q"], bad["r"], bad["beta"])


def test_julia_shim_matches_header():
    """pioran.jl_amd/julia/PioranHIP.jl cannot be executed here (no julia in the image): check every ccall statically
    against the prototypes of include/pioran_hip.h — symbol exists, same number of arguments, and each position has the
    same kind (32-bit int / 64-bit int / double / pointer), return type included."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    hdr = re.sub(r"/\*.*?\*/", "", (root / "include" / "pioran_hip.h").read_text(), flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"([\w \*]+?)\s*\b(pioran_\w+)\s*\(([^)]*)\)\s*;", hdr):
        def kind(a):
            a = a.strip()
            if "*" in a:
                return "ptr"
            if a.startswith("int64_t"):
                return "i64"
            if a.startswith("double"):
                return "f64"
            if a.startswith("int") or a.startswith("int32_t"):
                return "i32"
            if a == "void" or a == "":
                return None
            raise AssertionError(f"unknown C type {a!r} in {name}")
        ks = [kind(a) for a in args.split(",")]
        protos[name] = ([k for k in ks if k], "ptr" if "*" in ret else ("i32" if "int" in ret else ret.strip()))
    jl = (root / "pioran.jl_amd" / "julia" / "PioranHIP.jl").read_text()
    jkind = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cdouble": "f64", "Cstring": "ptr"}
    calls = re.findall(r"ccall\(\(:(\w+), LIB\),\s*(\w+),\s*\(([^)]*)\)", jl, flags=re.S)
    assert len(calls) >= 20
    seen = set()
    for name, ret, args in calls:
        assert name in protos, f"{name} is not declared in include/pioran_hip.h"
        seen.add(name)
        want, wret = protos[name]
        toks = [a.strip() for a in args.split(",") if a.strip()]
        got = [("ptr" if a.startswith(("Ptr{", "Ref{")) else jkind[a]) for a in toks]
        assert got == want, f"{name}: Julia passes {got}, the header declares {want}"
        assert ("ptr" if ret == "Cstring" else jkind[ret]) == wret, name
    # the entries a Pioran.jl maintainer needs are all bound
    for must in ("pioran_abi_version", "pioran_celerite_logl", "pioran_celerite_logl_batch", "pioran_celerite_logl_grad",
                 "pioran_logpdf_batch_theta", "pioran_dense_nll", "pioran_celerite_predict", "pioran_farm_logl_batch"):
        assert must in seen, must
    assert "pioran_abi_version" in jl and "ABI_VERSION = 7" in jl


def test_tile_dispatch_ladder_matches_the_committed_sweep():
    """capi.hip's automatic choice between the windowed form with one draw per wavefront and the other families (pioran_tile_choice: a pure
    function, no GPU) against the sweep it was tuned on (profiles/r05_tile_batch_sweep.txt, tools/ab_tile.py): on every measured (rows, batch
    size) the family the library takes is within 5 % of the faster one.  (the ladder was hand-typed from one box's sweep and
    nothing re-derived it; tools/retune_thresholds.py prints the table.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("retune_thresholds", ROOT / "tools" / "retune_thresholds.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    lines = list(mod.sweep_lines(ROOT / "profiles" / "r05_tile_batch_sweep.txt"))
    assert len(lines) >= 60
    bad = mod.check(verbose=False)
    assert not bad, [(r["model"], r["rows"], r["B"], round(loss, 3)) for r, loss in bad]


def test_bench_final_line_is_compact_and_ordered(capsys, tmp_path, monkeypatch):
    """bench.py's contract with the driver (which keeps an 8 KB tail of stdout): ONE final line of at most bench.LINE_LIMIT bytes whatever the size of the
    secondary measurements; the contract's keys, `roofline`, `cpu_baseline` and the parity figures always present; `secondary` in BASELINE's order (the other
    basis, dense, single evaluation, gradient, ...) with prose and counts dropped, cut from the END when too long; the long form in the file named by
    PIORAN_BENCH_FULL."""
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    import bench
    full = tmp_path / "full.json"
    monkeypatch.setenv("PIORAN_BENCH_FULL", str(full))
    big = {f"zz_section_{i}": {"note": "x" * 500, **{f"entry_{j}": {"time_ms": 1.0 + j, "kernel": "k" * 60, "count": j} for j in range(12)}} for i in range(30)}
    result = {"metric": "m", "value": 1.0, "unit": "evals/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 10.9, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w"}, "roofline": {"bound": "mfma", "frac": 0.45, "note": "n" * 3000},
              "cpu_baseline": {"value": 1.0, "unit": "evals/s", "cores": 32, "kind": "port", "sample": "s", "sample_detail": "d" * 2000},
              "max_rel_dlogl_vs_oracle": 1e-10, "max_abs_dlogl_vs_oracle_kept": 1e-8, "status_ok_frac": 0.9,
              "secondary": {**big, "gradient_sho20_N10000": {"chains_4096": {"value_and_gradient_ms": 50.0, "kernel": "tile"}},
                            "dense_n4096_j40": {"factor_ms": 1.13, "roofline": {"frac": 0.26, "peak": 78.6}},
                            "drwcelerite20_b4096": {"evals_per_s": 2.4e5, "roofline_frac_executed_rows": 0.63, "rows_executed": 60, "max_rel_dlogl_vs_oracle": 1e-11}}}
    bench.emit(result)
    line = capsys.readouterr().out.strip().splitlines()
    assert len(line) == 1 and len(line[0].encode()) <= bench.LINE_LIMIT
    d = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "max_rel_dlogl_vs_oracle", "max_abs_dlogl_vs_oracle_kept", "status_ok_frac", "other_basis"):
        assert k in d, k
    assert "note" not in d["roofline"] and "sample_detail" not in d["cpu_baseline"]
    keys = list(d["secondary"].keys())
    assert keys[:3] == ["drwcelerite20_b4096", "dense_n4096_j40", "gradient_sho20_N10000"] and d.get("secondary_truncated") is True
    assert d["secondary"]["dense_n4096_j40"] == {"factor_ms": 1.13, "roofline": {"frac": 0.26}}
    assert json.loads(full.read_text())["roofline"]["note"] == "n" * 3000
    (ROOT / "bench_full.json").unlink(missing_ok=True)
