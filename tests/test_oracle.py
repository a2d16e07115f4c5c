"""CPU tests (-m "not gpu"): pin the oracle (oracle/) against the reference's own outputs and literals.

The oracle is test infrastructure; these tests are what makes it trustworthy as the parity checker
for the HIP path (tests/test_gpu_parity.py).
"""
import json

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def lit(golden_dir):
    return json.loads((golden_dir / "reference_literals.json").read_text())


@pytest.fixture(scope="module")
def un(golden_dir):
    return np.load(golden_dir / "ultranest_points.npz")


def _un_inputs(un, i):
    t, y, yerr = un["t"], un["y"], un["yerr"]
    a1, f1, a2, var, nu, mu, cs = un["params"][i]
    f_min, f_max = 1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2
    a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, a1, f1, a2), f_min, f_max, 20, var,
                          is_integrated_power=False)
    return a, b, c, d, t, np.log(y - cs) - mu, nu * yerr ** 2 / (y - cs) ** 2


def test_reference_outputs_all_points(un):
    """Every log-likelihood the reference itself computed in its stored ultranest run
    (docs/src/data/inference, N=242, SHO-20 => J=20 terms) is reproduced by the C oracle."""
    logl = un["logl"]
    got = np.empty_like(logl)
    for i in range(len(logl)):
        got[i] = O.logl(*_un_inputs(un, i))
    rel = np.abs(got - logl) / np.abs(logl)
    assert len(logl) > 5000
    assert rel.max() < 1e-11, rel.max()


def test_numpy_twin_agrees(un):
    for i in (0, 17, 4242):
        args = _un_inputs(un, i)
        assert O.logl_numpy(*args) == pytest.approx(O.logl(*args), rel=1e-12)


def test_psd_amplitude_golden(lit):
    g = lit["psd_amplitudes"]
    amp = O.get_approx_coefficients(lambda f: O.single_bending_power_law(f, *g["params"]), g["f0"], g["fM"],
                                    n_components=g["J"])
    # reference uses isapprox (rtol sqrt(eps)); we hold 1e-9 elementwise (cond(B) ~ 2.6e3)
    np.testing.assert_allclose(amp, g["expected"], rtol=1e-9)


def test_coefficient_literals(lit):
    assert O.celerite_coefs_celerite(*lit["coefs_celerite"]["args"]) == lit["coefs_celerite"]["expected"]
    assert O.celerite_coefs_exp(*lit["coefs_exp"]["args"]) == lit["coefs_exp"]["expected"]
    s = lit["coefs_sho"]
    w0 = 2 * np.pi * s["w0_over_2pi"]
    assert O.celerite_coefs_sho(s["A"], w0, 1 / np.sqrt(2)) == [s["A"], s["A"], np.sqrt(2) / 2 * w0, np.sqrt(2) / 2 * w0]
    with pytest.raises(ValueError, match="not implemented yet"):
        O.celerite_coefs_sho(s["A"], w0, 0.5)
    g = lit["carma32"]
    r = [complex(*x) for x in g["r_alpha"]]
    a, b, c, d = O.carma_celerite_coefs(g["p"], r, g["beta"], g["norm"])
    for got, exp in zip((a, b, c, d), g["expected"]):
        np.testing.assert_allclose(got, exp, rtol=1.5e-8, atol=1e-15)


def test_relation_celerite_equals_dense(lit, golden_dir):
    """The reference's own known-answer relation (test/test_scalablegp.jl:128, test/test_likelihood.jl:58-59)."""
    rel = {c["name"]: c for c in json.loads((golden_dir / "relation_cases.json").read_text())["cases"]}
    g = lit["scalablegp_n6"]
    t = np.array(g["t"]); y = np.array(g["y"]); yerr = np.array(g["yerr"])
    for i in range(10):
        a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, g["alpha1"][i], g["f1"][i], g["alpha2"][i]),
                              g["f_min"], g["f_max"], g["n_components"], g["variance"][i])
        cel = O.logl(a, b, c, d, t, y - g["mu"][i], yerr ** 2)
        den = -O.dense_nll(a, b, c, d, t, y - g["mu"][i], yerr ** 2)
        assert np.isfinite(cel)
        assert cel == pytest.approx(den, rel=1e-11)
        assert cel == pytest.approx(rel[f"scalablegp_n6[{i}]"]["logl_mpmath50"], rel=1e-11)
    A = np.loadtxt(golden_dir / "simu_log.txt")
    t, y, yerr = A[:, 0], A[:, 1], A[:, 2]
    f0 = 1 / (t[-1] - t[0]) / 100
    fM = 1 / np.min(np.diff(t)) / 2 * 20
    for basis, J in (("SHO", 20), ("DRWCelerite", 40)):
        a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, 0.82, 0.01, 3.3), f0, fM, 20,
                              np.var(y, ddof=1), basis_function=basis)
        assert len(a) == J
        cel = O.logl(a, b, c, d, t, y, yerr ** 2)
        den = -O.dense_nll(a, b, c, d, t, y, yerr ** 2)
        assert cel == pytest.approx(den, rel=1e-11)
        assert cel == pytest.approx(O.dense_nll_numpy(a, b, c, d, t, y, yerr ** 2) * -1, rel=1e-11)
        assert cel == pytest.approx(rel[f"simu_log[{basis}]"]["logl_celerite"], rel=1e-13)


def test_kernel_closed_forms():
    """test/test_covariancefunctions.jl:3-30: k(tau) closed forms via the (a,b,c,d) representation."""
    tt = np.linspace(0, 10, 50)
    a, b, c, d = O.celerite_coefs_exp(1.0, 2.4)
    np.testing.assert_allclose([O.kappa([a], [b], [c], [d], x) for x in tt], np.exp(-tt * 2.4) / 2, rtol=1e-14)
    a, b, c, d = 1.3, 4.0, 0.5, 3.2
    np.testing.assert_allclose([O.kappa([a], [b], [c], [d], x) for x in tt],
                               np.exp(-c * tt) * (a * np.cos(d * tt) + b * np.sin(d * tt)), rtol=1e-13, atol=1e-15)


def test_r0_equals_variance():
    """test/test_psd.jl:100-153: with is_integrated_power=false, R(0) = sum(a) = variance."""
    for basis in ("SHO", "DRWCelerite"):
        a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, 0.2, 1.3e-2, 3.2), 2e-3, 3.52e2, 25, 1.32,
                              is_integrated_power=False, basis_function=basis)
        assert a.sum() == pytest.approx(1.32, rel=1e-12)
        assert len(a) == (25 if basis == "SHO" else 50)


def test_qpo_term_counts():
    """test/test_psd.jl:206-285: each QPO feature appends one celerite term."""
    a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, 0.2, 1.3e-2, 3.2), 2e-3, 3.52e2, 25, 1.32,
                          is_integrated_power=False, qpo_features=[(2.0, 1e-2, 14.2), (4.0, 1e-1, 4.2)])
    assert len(a) == 27
    a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, 0.2, 1.3e-2, 4.2), 2e-3, 3.52e2, 25, 1.32,
                          is_integrated_power=False, basis_function="DRWCelerite", qpo_features=[(1.4, 1e-2, 10.2)])
    assert len(a) == 51


def test_status_and_batch():
    rng = np.random.default_rng(1)
    t = np.cumsum(rng.uniform(0.1, 2, 50)); y = rng.standard_normal(50); s2 = np.full(50, 0.01)
    A = rng.uniform(0.1, 1, (5, 3)); Bc = rng.uniform(0, 0.1, (5, 3))
    C = rng.uniform(0.1, 1, 3); Dd = rng.uniform(0.1, 2, 3)
    mu = rng.standard_normal(5); nu = rng.uniform(0.5, 2, 5)
    out, st = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=2, return_status=True)
    for i in range(5):
        assert out[i] == O.logl(A[i], Bc[i], C, Dd, t, y - mu[i], nu[i] * s2)
    assert (st == 0).all()
    # non positive-definite: negative amplitude with tiny noise -> status 1 (D_n <= 0)
    v, s = O.logl([-5.0], [0.0], [0.1], [0.0], t, y, s2 * 1e-6, return_status=True)
    assert s != 0


def test_sim_matches_covariance():
    """sim (src/celerite_solver.jl:515-549) draws have covariance K: check L-factor identity y = L q."""
    rng = np.random.default_rng(3)
    t = np.cumsum(rng.uniform(0.1, 1, 12)); s2 = np.full(12, 0.05)
    a, b, c, d = [1.0, 0.4], [0.2, 0.0], [0.3, 1.0], [1.1, 0.0]
    # columns of the implied factor: y(q = e_k)
    Lf = np.stack([O.sim(a, b, c, d, t, s2, np.eye(12)[k]) for k in range(12)], axis=1)
    K = np.array([[O.kappa(a, b, c, d, abs(ti - tj)) for tj in t] for ti in t]) + np.diag(s2)
    np.testing.assert_allclose(Lf @ Lf.T, K, rtol=1e-11, atol=1e-13)


def test_predict_equals_dense_prediction(golden_dir):
    """The reference pins its celerite `predict` against the dense `predict_direct` (isapprox, rtol 1.5e-8):
    test/test_scalablegp.jl:134-157 (N = 6 literals, tau on / between / outside the data),
    test/test_prediction.jl:49-58 and test/test_predict_celerite.jl:3-28 (test/data/simu.txt, N = 489)."""
    t = np.array([0.0, 3.0, 3.2, 3.4, 45.5, 101.2])
    tx = np.array([0.0, 1.4, 2.3, 3.0, 3.1, 3.2, 3.3, 3.4, 45.5, 101.2, 202.32])
    y = np.array([1.3, 2.2, 4.21, 2.5, 3.3, 5.2]); yerr = np.array([0.1, 0.2, 0.1, 0.1, 0.2, 0.1])
    a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, 0.2, 0.02, 3.1), 1e-4, 1e1, 30, 2.31)
    for tau in (t, tx):
        np.testing.assert_allclose(O.predict(a, b, c, d, tau, t, y - 1.2, yerr ** 2),
                                   O.predict_direct_numpy(a, b, c, d, tau, t, y - 1.2, yerr ** 2), rtol=1e-10, atol=1e-12)
    A = np.loadtxt(golden_dir / "simu.txt")
    t, y, yerr = A[:, 0], A[:, 1], A[:, 2]
    f0, fM = 1 / (t[-1] - t[0]) / 100, 1 / np.min(np.diff(t)) / 2 * 20
    rng = np.random.default_rng(0)
    grids = (t, np.linspace(t.min(), t.max(), 1000), np.linspace(t.min() - 30, t.max() + 30, 1000),
             np.sort(rng.random(1000)) * (t[-1] - t[0]) * 2 + (t[0] - t[-1] / 2))
    coefs = [O.approx(lambda f: O.single_bending_power_law(f, 0.82, 0.01, 3.3), f0, fM, 20, np.var(y, ddof=1)),
             ([0.5], [0.0], [2.4], [0.0]),          # Exp(1.0, 2.4): a = A / 2 (src/Exp.jl:29-33)
             ([3.2], [0.2], [3.0], [0.2])]          # Celerite(3.2, 0.2, 3.0, 0.2)
    for a, b, c, d in coefs:
        for tau in grids:
            np.testing.assert_allclose(O.predict(a, b, c, d, tau, t, y, yerr ** 2),
                                       O.predict_direct_numpy(a, b, c, d, tau, t, y, yerr ** 2), rtol=1e-9, atol=1e-11)


def test_complex_step_twin_matches_logl_and_finite_differences():
    """The complex twin of the restatement (oracle/celerite_oracle_cstep.c) returns the same log L and derivatives that
    agree with central differences — it is the reference for the device gradient."""
    rng = np.random.default_rng(0)
    N, J = 40, 3
    t = np.cumsum(rng.uniform(0.1, 1.5, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
    a = rng.uniform(0.3, 1.5, J); b = rng.uniform(-0.05, 0.05, J) * a; c = rng.uniform(0.1, 1.0, J); d = rng.uniform(0.2, 2.0, J)
    g = O.logl_grad(a, b, c, d, t, y, s2, series=True)
    h = 1e-6
    E = np.eye(J)
    fa = np.array([(O.logl(a + h * E[j], b, c, d, t, y, s2) - O.logl(a - h * E[j], b, c, d, t, y, s2)) / (2 * h) for j in range(J)])
    fb = np.array([(O.logl(a, b + h * E[j], c, d, t, y, s2) - O.logl(a, b - h * E[j], c, d, t, y, s2)) / (2 * h) for j in range(J)])
    np.testing.assert_allclose(g["grad_a"], fa, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(g["grad_b"], fb, rtol=1e-6, atol=1e-7)
    En = np.eye(N)
    fy = np.array([(O.logl(a, b, c, d, t, y + h * En[k], s2) - O.logl(a, b, c, d, t, y - h * En[k], s2)) / (2 * h) for k in range(N)])
    np.testing.assert_allclose(g["grad_y"], fy, rtol=1e-6, atol=1e-7)
    # directional form: d/dmu = -sum_n d/dy_n, d/dnu = sum_n sigma2_n d/d(sigma2_n)
    assert abs(O.logl_dir(a, b, c, d, t, y, s2, dy=-np.ones(N)) + g["grad_y"].sum()) < 1e-9 * N
    assert abs(O.logl_dir(a, b, c, d, t, y, s2, ds2=s2) - g["grad_sigma2"] @ s2) < 1e-9 * N


# ---- the three further stored runs of the reference (examples/ultranest/inference/*) ---------------------------------
@pytest.fixture(scope="module")
def runs(golden_dir):
    return np.load(golden_dir / "ultranest_example_runs.npz")


def example_run_inputs(runs, name, rows):
    """(A, Bc, C, Dd, t, y, s2, mu, nu, Y) of the model each run was made with (oracle/make_golden.py:
    make_example_runs_fixture): Y is None except for the CustomMean run, where it is the per-draw y - mean(t)."""
    t, y, yerr, P = (runs[f"{name}_{k}"] for k in ("t", "y", "yerr", "params"))
    P = P[rows]
    f_min, f_max = 1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2
    A, Bc = np.empty((len(P), 20)), np.empty((len(P), 20))
    for i, p in enumerate(P):
        if name == "simu_double":
            psd = lambda f, p=p: O.double_bending_power_law(f, *p[:5])  # noqa: E731
            var = p[5]
        else:
            psd = lambda f, p=p: O.single_bending_power_law(f, *p[:3])  # noqa: E731
            var = p[3]
        A[i], Bc[i], C, Dd = O.approx(psd, f_min, f_max, 20, var)
    if name == "simu_periodic":
        amp, ph, T0 = P[:, 6:7], P[:, 7:8], P[:, 8:9]
        Y = y[None, :] - amp * np.sin(2 * np.pi * t[None, :] / T0 + ph)   # CustomMean minus its constant part mu
        return A, Bc, C, Dd, t, y, yerr ** 2, P[:, 5], P[:, 4], Y
    k = 6 if name == "simu_double" else 4
    return A, Bc, C, Dd, t, np.log(y), yerr ** 2 / y ** 2, P[:, k + 1], P[:, k], None


@pytest.mark.parametrize("name", ["simu_single", "simu_double", "simu_periodic"])
def test_reference_outputs_example_runs(runs, name):
    """Every log-likelihood the reference computed in its three other stored runs — integrated-power normalisation,
    DoubleBendingPowerLaw, a CustomMean sinusoid (examples/ultranest/*.jl) — is reproduced by the C oracle.
    Median error 4e-15, 99.9 % of the points < 3e-11; the tail (8 of 20 297 points between 1e-10 and 4.4e-10) are prior
    draws with alpha_2 -> 4, the SHO basis' limit, where approx's spectral solve is ill-conditioned and Julia's LU and
    LAPACK's round differently — still 20x inside the 1e-8 north-star bar."""
    logl = runs[f"{name}_logl"]
    assert len(logl) > 6000
    A, Bc, C, Dd, t, y, s2, mu, nu, Y = example_run_inputs(runs, name, slice(None))
    if Y is None:
        got = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu, nthreads=8)
    else:
        got = np.array([O.logl(A[i], Bc[i], C, Dd, t, Y[i] - mu[i], nu[i] * s2) for i in range(len(logl))])
    rel = np.abs(got - logl) / np.abs(logl)
    assert np.median(rel) < 1e-13
    assert np.quantile(rel, 0.999) < 1e-10
    assert rel.max() < 1e-9, rel.max()


# ---- the reference's stored NUTS chains (docs/src/data/subset_simu_single.h5): log_density - log prior - log |Jacobian| ----------
def test_reference_outputs_turing_chain(golden_dir):
    """8183 distinct draws of the reference's own NUTS run (docs/src/turing.md:170-256; N = 250, SHO-20, sampled shift): the
    log-likelihood inside the `log_density` the sampler stored — high-likelihood region — is reproduced by the C oracle.
    (The fixture's decomposition is closed-form in theta: oracle/make_golden.py make_turing_chain_fixture.)"""
    tc = np.load(golden_dir / "turing_chain.npz")
    t, y, yerr, P, ref = tc["t"], tc["y"], tc["yerr"], tc["params"], tc["logl"]
    assert len(ref) > 8000 and np.allclose(tc["log_density"] - tc["log_prior_plus_log_jacobian"], ref, rtol=0, atol=1e-11)
    f_min, f_max = 1 / (t[-1] - t[0]), 1 / np.min(np.diff(t)) / 2
    got = np.empty_like(ref)
    for i, (a1, f1, a2, var, nu, mu, cs) in enumerate(P):
        a, b, c, d = O.approx(lambda f: O.single_bending_power_law(f, a1, f1, a2), f_min, f_max, 20, var, is_integrated_power=False)
        got[i] = O.logl(a, b, c, d, t, np.log(y - cs) - mu, nu * yerr ** 2 / (y - cs) ** 2)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    # the stored draws are the constrained images of the sampler's unconstrained points: one ulp of that round trip times the
    # posterior's curvature is all that separates the two evaluations
    assert err.max() < 1e-10, (err.max(), int(np.argmax(err)))
    assert (ref > ref.max() - 30).sum() > 6000          # most of them sit where a sampler lives


# ---- ill-conditioned draws: the fp64 oracle against the same recurrence in __float128 (oracle/celerite_oracle_q.c) -------------------
def test_quad_truth_fixture_and_fp64_oracle(golden_dir):
    """tests/golden/quad_truth.npz (oracle/make_quad_truth.py): prior draws of the bench model with ratio = nu min(sigma2) / sum(a) ~
    1 / cond(K) down to 3e-10, log L evaluated in quad precision.  (1) The quad code reproduces the stored values (a few draws at N = 150:
    libquadmath is software arithmetic); (2) the fp64 oracle — the reference's algorithm and operation order — is itself up to 8e-9 from the
    exact value of its fp64 inputs below ratio 1e-8 (rounding sum(a), nu sigma2 and the phases d t alone moves log L by ~ eps / ratio there:
    tools/window_precision_study.py), and within 2e-9 from ratio 1e-8 on.  The GPU families are held against the same truth in
    tests/test_gpu_parity.py::test_ill_conditioned_draws_vs_quad_truth."""
    q = np.load(golden_dir / "quad_truth.npz")
    t, y, yerr = q["n150_t"], q["n150_y"], q["n150_yerr"]
    A, Bc, C, Dd, mu, nu = (q[f"n150_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
    pick = np.argsort(q["n150_ratio"])[[0, 3, 40, 200]]
    again = O.logl_quad_batch(A[pick], Bc[pick], C, Dd, t, y, yerr ** 2, mu[pick], nu[pick], nthreads=4)
    assert np.array_equal(again, q["n150_truth"][pick])
    for tag, series in (("n150", (t, y, yerr)), ("n1000", (q["n1000_t"], q["n1000_y"], q["n1000_yerr"]))):
        tt, yy, ee = series
        A, Bc, C, Dd, mu, nu = (q[f"{tag}_{k}"] for k in ("A", "Bc", "C", "Dd", "mu", "nu"))
        truth, ratio = q[f"{tag}_truth"], q[f"{tag}_ratio"]
        got, st = O.logl_batch(A, Bc, C, Dd, tt, yy, ee ** 2, mu, nu, nthreads=8, return_status=True)
        assert (st == 0).all() and (q[f"{tag}_dmin"] > 0).all()
        err = np.abs(got - truth) / np.abs(truth)
        assert err[ratio >= 1e-8].max() < 2e-9 and err[ratio < 1e-8].max() < 1e-8, (tag, err[ratio >= 1e-8].max(), err[ratio < 1e-8].max())
        assert np.median(err) < 1e-11
