"""GPU test (-m gpu): one larger run of the prediction / simulation / batched dense paths — 601 draws (three chunks of the
entries' 256), N = 5003 (not a multiple of the 16-step window or of the 128-step segments), M = 2999 evaluation times ascending
and permuted — against the oracle's `pred` / `sim` (src/celerite_solver.jl:363-483, 515-549) and against each other; a batched
dense launch of 40 matrices at N = 3000 (steps in fours) against single calls.  (tools/big_check.py of round 3, as a test.)"""
import numpy as np
import pytest

import bench
import pioran_jl_amd as pj
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case():
    N, J, B, M = 5003, 20, 601, 2999
    t, y, yerr = bench.synth_series(10_000)
    t, y, yerr = t[:N], y[:N], yerr[:N]
    th, _, _ = bench.synth_theta(B, t, y, seed=99)
    f_min, f_max = 1 / (t[-1] - t[0]), 1 / (2 * np.min(np.diff(t)))
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
    ctx = pj.Context(0)
    return dict(N=N, J=J, B=B, M=M, t=t, y=y, yerr=yerr, A=A, Bc=Bc, C=C, Dd=Dd, mu=th[:, 5].copy(), nu=th[:, 4].copy(), ctx=ctx,
                ds=pj.Dataset(t, y, yerr ** 2, ctx), tau=np.sort(np.random.default_rng(1).uniform(t[0] - 5, t[-1] + 5, M)))


def test_predict_601_draws_three_chunks(case):
    c = case
    got, st = c["ds"].predict(c["A"], c["Bc"], c["C"], c["Dd"], c["tau"], mu=c["mu"], nu=c["nu"], return_status=True)
    assert pj._lib.lib().pioran_celerite_config_name(-1).decode() == "block (windowed prediction)"
    ok = np.flatnonzero(st == 0)
    assert len(ok) > 550
    for i in (ok[0], ok[len(ok) // 2], ok[-1], ok[255], ok[256], ok[512]):         # both sides of every chunk boundary
        ref = O.predict(c["A"][i], c["Bc"][i], c["C"], c["Dd"], c["tau"], c["t"], c["y"] - c["mu"][i], c["nu"][i] * c["yerr"] ** 2) + c["mu"][i]
        assert np.max(np.abs(got[i] - ref)) / np.max(np.abs(ref)) < 1e-10
    # permuted evaluation times: the segment passes + the 16-lanes-per-time evaluation instead of the fused kernel.  Both form, per
    # evaluation time, the same sum of 2 R products Q_r(n0) e^{-c_r dt} (a_r cos + b_r sin)(d_r tau) in a different order — terms that are
    # individually 1e2 .. 1e4 times the predicted value and cancel — so the two paths differ by rounding RELATIVE TO THOSE TERMS: ~1e-11 of
    # the curve's scale, the same size as each path's own distance from the oracle (2.5e-12 above).  Round 3's "unsorted vs sorted 1.4e-9"
    # (profiles/r03_big_check.txt) was the ELEMENTWISE ratio of that 2e-11 at a time where the curve itself is ~1e-2: printed, then bounded
    # for what it is.
    perm = np.random.default_rng(3).permutation(c["M"])
    g2 = c["ds"].predict(c["A"][:5], c["Bc"][:5], c["C"], c["Dd"], c["tau"][perm], mu=c["mu"][:5], nu=c["nu"][:5])
    g1 = got[:5][:, perm]
    okd = np.isfinite(g1).all(axis=1)
    diff = np.abs(g2[okd] - g1[okd])
    scale = np.max(np.abs(g1[okd] - c["mu"][:5][okd, None]), axis=1, keepdims=True)   # the mean function is added last
    k = np.unravel_index(np.argmax(diff / np.maximum(np.abs(g1[okd]), 1e-300)), diff.shape)
    kd = np.unravel_index(np.argmax(diff / scale), diff.shape)
    d = int(np.flatnonzero(okd)[kd[0]])
    # (the oracle walks the merged (t, tau) sequence like the reference, src/celerite_solver.jl:392-479: ascending tau, then permuted)
    ref = (O.predict(c["A"][d], c["Bc"][d], c["C"], c["Dd"], c["tau"], c["t"], c["y"] - c["mu"][d], c["nu"][d] * c["yerr"] ** 2) + c["mu"][d])[perm]
    print(f"unsorted vs sorted: max |diff| / curve scale = {np.max(diff / scale):.2e} (draw {d}, time index {kd[1]}: fused - oracle "
          f"{(g1[okd][kd] - ref[kd[1]]) / scale[kd[0], 0]:.2e}, two-kernel - oracle {(g2[okd][kd] - ref[kd[1]]) / scale[kd[0], 0]:.2e} of the scale); "
          f"worst elementwise ratio {diff[k] / abs(g1[okd][k]):.2e} at |value| / scale = {abs(g1[okd][k]) / scale[k[0], 0]:.2e}")
    assert np.max(diff / scale) < 1e-10
    assert np.max(np.abs(g2[kd[0]] - ref)) / np.max(np.abs(ref)) < 1e-10


def test_simulate_601_draws(case):
    c = case
    q = np.random.default_rng(2).standard_normal((c["B"], c["N"]))
    ys = c["ctx"].simulate(c["A"], c["Bc"], c["C"], c["Dd"], c["t"], c["yerr"] ** 2, q)
    assert pj._lib.lib().pioran_celerite_config_name(-1).decode() == "block (windowed simulation)"
    fin = np.flatnonzero(np.isfinite(ys).all(axis=1))
    assert len(fin) > 550
    for i in (fin[0], fin[-1], fin[256], fin[300]):
        ref = O.sim(c["A"][i], c["Bc"][i], c["C"], c["Dd"], c["t"], c["yerr"] ** 2, q[i])
        assert np.max(np.abs(ys[i] - ref)) / np.max(np.abs(ref)) < 1e-10


def test_dense_batch_of_40_equals_single_calls(case):
    c = case
    n = 3000
    t, y, s2 = c["t"][:n], c["y"][:n], c["yerr"][:n] ** 2 + 1.0
    v = c["ctx"].dense_nll_batch(c["A"][:40] / 50, c["Bc"][:40] / 50, c["C"], c["Dd"], t, y, s2, mu=c["mu"][:40])
    for i in (0, 17, 39):
        one = c["ctx"].dense_nll(c["A"][i] / 50, c["Bc"][i] / 50, c["C"], c["Dd"], t, y - c["mu"][i], s2)
        assert np.isfinite(one) and abs(v[i] - one) <= 1e-13 * abs(one)
    ref = O.dense_nll(c["A"][17] / 50, c["Bc"][17] / 50, c["C"], c["Dd"], t, y - c["mu"][17], s2)
    assert abs(v[17] - ref) <= 1e-10 * abs(ref)
