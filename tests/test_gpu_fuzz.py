"""GPU (-m gpu): a seeded, case-bounded slice of tools/fuzz_layouts.py — the windowed kernel (celerite_block.hip) and the
throughput scan in its two-step form (celerite_scan.hip) against the CPU oracle on random shapes: 1..31 terms (some of them
one-row terms), N = 1..699 with occasional long gaps, 1..39 draws, optional mu / nu / per-draw series.

Same generator, same seed as the tool, cases 0..1499 (every case is a function of (seed, index) alone).  The bar is the
north-star's 1e-8 relative to max(1, |log L|); the test PRINTS the worst case it saw (index, J, N, B, one-row terms, layout,
draw) so that an outlier can be regenerated with tools/explain_outliers.py --case <index>.
"""
import importlib.util
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
NCASES = 1500


def _fuzz_module():
    spec = importlib.util.spec_from_file_location("_fuzz_layouts", ROOT / "tools" / "fuzz_layouts.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fuzz_slice_block_and_scan_vs_oracle(capsys):
    import pioran_jl_amd as pj
    from oracle import oracle as O
    fz = _fuzz_module()
    s = fz.fuzz(pj, O, pj.Context(0), seed=fz.SEED, ncases=NCASES, keep=3)
    with capsys.disabled():
        print(f"\nfuzz slice: {s['cases']} cases in {s['seconds']:.1f} s, worst deviation {s['worst_dev']:.2e}", file=sys.stderr)
        for r in s["worst"]:
            print(f"  case {r['idx']}: J={r['J']} N={r['N']} B={r['B']} nreal={r['nreal']} per-draw-series={r['useY']} layout={r['layout']} "
                  f"draw={r['draw']} dev={r['dev']:.2e} (got {r['got']!r}, oracle {r['oracle']!r})", file=sys.stderr)
    assert s["cases"] == NCASES
    assert not s["failures"], s["failures"][:5]
    assert s["worst_dev"] < 1e-8


def test_fuzz_slice_32_to_47_terms_block_and_scan_vs_oracle(capsys):
    """The same generator with 32..47 terms (64..94 rows less the one-row terms): the windowed kernel's five / six block columns
    (round 4) and the throughput scan's 80- and 96-row shapes, 400 cases."""
    import pioran_jl_amd as pj
    from oracle import oracle as O
    fz = _fuzz_module()
    s = fz.fuzz(pj, O, pj.Context(0), seed=fz.SEED, ncases=400, keep=3, jrange=(32, 48))
    with capsys.disabled():
        print(f"\nfuzz slice, 32..47 terms: {s['cases']} cases in {s['seconds']:.1f} s, worst deviation {s['worst_dev']:.2e}", file=sys.stderr)
        for r in s["worst"]:
            print(f"  case {r['idx']}: J={r['J']} N={r['N']} B={r['B']} nreal={r['nreal']} per-draw-series={r['useY']} layout={r['layout']} "
                  f"draw={r['draw']} dev={r['dev']:.2e}", file=sys.stderr)
    assert s["cases"] == 400
    assert not s["failures"], s["failures"][:5]
    assert s["worst_dev"] < 1e-8


def test_fuzz_windowed_gradient_vs_step_by_step_and_oracle(capsys):
    """Randomized shapes for the windowed reverse mode (celerite_block_adjoint_kernel): 300 seeded cases — 3..31 terms (every seventh case: one or two), some of them
    one-row terms, N = 1..400 with occasional long gaps, 1..4 chains, shared or per-draw (c, d) — every gradient component against
    the step-by-step adjoint kernels (a different algorithm on a different kernel), every tenth case against the complex-step oracle."""
    import numpy as np
    import pioran_jl_amd as pj
    from oracle import oracle as O
    ctx = pj.Context(0)
    name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
    worst = {"dev": 0.0}
    keys = ("grad_a", "grad_b", "grad_c", "grad_d", "grad_mu", "grad_nu")
    for idx in range(300):
        rng = np.random.default_rng([20261004, idx])
        J = int(rng.integers(3, 32)); N = int(rng.integers(1, 401)); B = int(rng.integers(1, 5))
        if idx % 7 == 3:
            J = 1 + idx % 2          # one and two terms (1 .. 4 rows)
        gaps = rng.uniform(0.05, 2.0, N)
        if rng.random() < 0.3:
            gaps[rng.integers(0, N, max(1, N // 20))] *= rng.uniform(5, 400)
        t = np.cumsum(gaps); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        per_draw = bool(rng.random() < 0.35) and B > 1
        C = rng.uniform(0.05, 2.0, (B, J) if per_draw else (J,)); Dd = rng.uniform(0.0, 3.0, (B, J) if per_draw else (J,))
        nreal = 0 if per_draw else int(rng.integers(0, J // 2 + 1)) * int(rng.random() < 0.4)
        Bc[:, :nreal] = 0.0
        if not per_draw:
            Dd[:nreal] = 0.0
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        ds = pj.Dataset(t, y, s2, ctx)
        gw = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
        R = 2 * J - nreal
        want = "block (windowed gradient, per-draw tables)" if per_draw else "block (windowed gradient)"
        assert name() == want, (idx, name(), R)     # (fewer than six rows too, since late round 4: 16.4 -> 4.4 ms at N = 1e4)
        try:
            ctx.set_option("no_block", True)
            go = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
        finally:
            ctx.set_option("no_block", False)
        ok = (gw["status"] == 0) & (go["status"] == 0)
        assert np.array_equal(gw["status"], go["status"]), idx
        for k in keys:
            if not ok.any():
                continue
            dev = float(np.max(np.abs(gw[k][ok] - go[k][ok])) / (1 + np.max(np.abs(go[k][ok]))))
            if dev > worst["dev"]:
                worst = {"dev": dev, "idx": idx, "key": k, "J": J, "N": N, "B": B, "nreal": nreal, "per_draw": per_draw}
        if idx % 10 == 0 and ok[0]:
            cd0 = (C[0], Dd[0]) if per_draw else (C, Dd)
            ref = O.logl_grad(A[0], Bc[0], cd0[0], cd0[1], t, y - mu[0], nu[0] * s2, cd=True)
            for k in ("grad_a", "grad_b", "grad_c", "grad_d"):
                assert np.max(np.abs(gw[k][0] - ref[k])) <= 1e-8 * (1 + np.max(np.abs(ref[k]))), (idx, k)
        ds.close()
    with capsys.disabled():
        print(f"\nwindowed gradient fuzz: 300 cases, worst deviation from the step-by-step adjoint {worst}", file=sys.stderr)
    assert worst["dev"] < 1e-8, worst


def test_fuzz_windowed_prediction_and_simulation_vs_step_by_step_and_oracle(capsys):
    """Randomized shapes for the windowed prediction (block back-substitution + segment-parallel running vectors) and simulation
    (L applied window by window): 200 seeded cases — 3..31 terms, some one-row terms, N = 1..700 (segment and window edges: 16, 128,
    129 ...) with occasional long gaps, 1..5 draws, tau inside / outside / exactly on the data — against the step-by-step kernels, every
    tenth case against the oracle's `pred` and `sim`."""
    import numpy as np
    import pioran_jl_amd as pj
    from oracle import oracle as O
    ctx = pj.Context(0)
    name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
    worst = {"dev": 0.0}
    edges = (1, 2, 15, 16, 17, 127, 128, 129, 256, 257)
    for idx in range(200):
        rng = np.random.default_rng([20261005, idx])
        J = int(rng.integers(3, 32)); B = int(rng.integers(1, 6))
        if idx % 7 == 3:
            J = 1 + idx % 2          # one and two terms (1 .. 4 rows: windowed too since late round 4)
        N = int(edges[idx % len(edges)]) if idx < 40 else int(rng.integers(1, 701))
        gaps = rng.uniform(0.05, 2.0, N)
        if rng.random() < 0.3:
            gaps[rng.integers(0, N, max(1, N // 20))] *= rng.uniform(5, 400)
        t = np.cumsum(gaps); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        nreal = int(rng.integers(0, J // 2 + 1)) * int(rng.random() < 0.4)
        Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        M = int(rng.integers(1, 200))
        tau = rng.uniform(t[0] - 3, t[-1] + 3, M); tau[: min(M, 3)] = t[rng.integers(0, N, min(M, 3))]
        tau = np.sort(tau)
        q = rng.standard_normal((B, N))
        R = 2 * J - nreal
        ds = pj.Dataset(t, y, s2, ctx)
        pw, st = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu, return_status=True)
        assert name() == "block (windowed prediction)", (idx, name(), R)
        sw = ctx.simulate(A, Bc, C, Dd, t, s2, q)
        assert name() == "block (windowed simulation)", (idx, name(), R)
        try:
            ctx.set_option("no_block", True)
            po, st2 = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu, return_status=True)
            so = ctx.simulate(A, Bc, C, Dd, t, s2, q)
        finally:
            ctx.set_option("no_block", False)
        assert np.array_equal(st, st2), idx
        ok = (st == 0)
        for k, a_, b_ in (("predict", pw, po), ("simulate", sw, so)):
            if not ok.any():
                continue
            dev = float(np.max(np.abs(a_[ok] - b_[ok]) / (1e-30 + np.max(np.abs(b_[ok]), axis=1, keepdims=True))))
            if dev > worst["dev"]:
                worst = {"dev": dev, "idx": idx, "what": k, "J": J, "N": N, "B": B, "nreal": nreal, "M": M}
        if idx % 10 == 0 and ok[0]:
            ref = O.predict(A[0], Bc[0], C, Dd, tau, t, y - mu[0], nu[0] * s2) + mu[0]
            assert np.max(np.abs(pw[0] - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), idx
            ref = O.sim(A[0], Bc[0], C, Dd, t, s2, q[0])
            assert np.max(np.abs(sw[0] - ref)) <= 1e-9 * np.max(np.abs(ref)), idx
        ds.close()
    with capsys.disabled():
        print(f"\nwindowed prediction / simulation fuzz: 200 cases, worst deviation from the step-by-step kernels {worst}", file=sys.stderr)
    assert worst["dev"] < 1e-8, worst


def test_fuzz_per_draw_prediction_simulation_and_dense_batches(capsys):
    """Randomized shapes for the paths added last in round 3: (a) prediction / simulation with (c, d) per draw, all draws in one launch
    (per-draw windowed tables), against the draw-by-draw step-by-step kernels — 80 cases; (b) batched dense launches (matrix index as a
    grid dimension, steps in fours / pairs, shared-(c, d) batch build on the matrix cores; unsorted time stamps and more than 64 terms
    take the older build kernels) against one matrix per call — 40 cases, every fifth against the LAPACK twin of the oracle."""
    import numpy as np
    import pioran_jl_amd as pj
    from oracle import oracle as O
    ctx = pj.Context(0)
    name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
    worst = {"dev": 0.0}
    for idx in range(80):
        rng = np.random.default_rng([20261006, idx])
        J = int(rng.integers(3, 32)); B = int(rng.integers(2, 7)); N = int(rng.integers(1, 500))
        gaps = rng.uniform(0.05, 2.0, N)
        if rng.random() < 0.3:
            gaps[rng.integers(0, N, max(1, N // 20))] *= rng.uniform(5, 200)
        t = np.cumsum(gaps); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, (B, J)); Dd = rng.uniform(0.0, 3.0, (B, J))
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        M = int(rng.integers(1, 150))
        tau = np.sort(rng.uniform(t[0] - 3, t[-1] + 3, M))
        q = rng.standard_normal((B, N))
        ds = pj.Dataset(t, y, s2, ctx)
        pw, st = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu, return_status=True)
        assert name() == "block (windowed prediction, per-draw tables)", (idx, name())
        sw = ctx.simulate(A, Bc, C, Dd, t, s2, q)
        assert name() == "block (windowed simulation, per-draw tables)", (idx, name())
        try:
            ctx.set_option("no_block", True)
            po, st2 = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu, return_status=True)
            so = ctx.simulate(A, Bc, C, Dd, t, s2, q)
        finally:
            ctx.set_option("no_block", False)
        assert np.array_equal(st, st2), idx
        ok = st == 0
        for k, a_, b_ in (("predict", pw, po), ("simulate", sw, so)):
            if ok.any():
                dev = float(np.max(np.abs(a_[ok] - b_[ok]) / (1e-30 + np.max(np.abs(b_[ok]), axis=1, keepdims=True))))
                if dev > worst["dev"]:
                    worst = {"dev": dev, "idx": idx, "what": k + " (per-draw c, d)", "J": J, "N": N, "B": B}
        ds.close()
    for idx in range(40):
        rng = np.random.default_rng([20261007, idx])
        J = int(rng.integers(1, 9)) if idx % 8 else 70        # (more than 64 terms: the older build kernels)
        B = int(rng.integers(2, 9)); N = int(rng.integers(1, 900)) if idx % 4 else int(rng.integers(1100, 1700))
        t = np.cumsum(rng.uniform(0.05, 2.0, N))
        if idx % 7 == 3:
            t = rng.permutation(t)                              # unsorted: direct evaluation of every entry
        y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)) / J; Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        per_draw = bool(rng.random() < 0.3)
        C = rng.uniform(0.05, 2.0, (B, J) if per_draw else (J,)); Dd = rng.uniform(0.0, 3.0, (B, J) if per_draw else (J,))
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        got, info = ctx.dense_nll_batch(A, Bc, C, Dd, t, y, s2, mu=mu, nu=nu, return_info=True)
        one = np.array([ctx.dense_nll(A[i], Bc[i], C[i] if per_draw else C, Dd[i] if per_draw else Dd, t, y - mu[i], nu[i] * s2) for i in range(B)])
        assert (info == 0).all(), idx
        dev = float(np.max(np.abs(got - one) / np.abs(one)))
        if dev > worst["dev"]:
            worst = {"dev": dev, "idx": idx, "what": "dense batch", "J": J, "N": N, "B": B, "per_draw": per_draw}
        if idx % 5 == 0:
            i = B - 1
            ref = O.dense_nll_numpy(A[i], Bc[i], C[i] if per_draw else C, Dd[i] if per_draw else Dd, t, y - mu[i], nu[i] * s2)
            assert abs(got[i] - ref) <= 1e-10 * abs(ref), (idx, got[i], ref)
    with capsys.disabled():
        print(f"\nper-draw prediction / simulation and dense batch fuzz: 120 cases, worst deviation from the one-at-a-time paths {worst}", file=sys.stderr)
    assert worst["dev"] < 1e-8, worst


def test_fuzz_64_to_143_rows_gradient_prediction_simulation_vs_oracle(capsys):
    """Randomized shapes past the windowed kernels (round 4: lean latency kernel with stores, lean reverse pass): 40 seeded cases —
    32..71 terms, some of them one-row terms (64..142 rows), N = 1..300 over one and several checkpoint segments, 1..3 draws —
    value + gradient against the complex-step oracle, posterior mean and simulation against the oracle's `pred` / `sim`."""
    import numpy as np
    import pioran_jl_amd as pj
    from oracle import oracle as O
    ctx = pj.Context(0)
    name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
    worst = {"grad": 0.0, "pred": 0.0, "sim": 0.0}
    edges = (1, 2, 15, 16, 17, 31, 32, 33, 48, 65)
    for idx in range(40):
        rng = np.random.default_rng([20261006, idx])
        J = int(rng.integers(32, 72)); B = int(rng.integers(1, 4))
        N = int(edges[idx]) if idx < len(edges) else int(rng.integers(1, 301))
        nreal = int(rng.integers(0, J // 3)) * int(rng.random() < 0.4)
        while 2 * J - nreal < 64:
            nreal -= 1
        R = 2 * J - nreal
        gaps = rng.uniform(0.05, 2.0, N)
        if rng.random() < 0.3:
            gaps[rng.integers(0, N, max(1, N // 20))] *= rng.uniform(5, 400)
        t = np.cumsum(gaps); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)) / J; Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
        ds = pj.Dataset(t, y, s2, ctx)
        g = ds.logl_grad(A, Bc, C, Dd, mu=mu, nu=nu)
        assert name() == "wide (step-by-step gradient)" and (g["status"] == 0).all(), (idx, name(), R)
        ref_l = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu)
        assert np.max(np.abs(g["logl"] - ref_l) / np.maximum(1.0, np.abs(ref_l))) < 1e-10, idx
        ref = O.logl_grad(A[0], Bc[0], C, Dd, t, y - mu[0], nu[0] * s2, cd=True)
        for k in ("grad_a", "grad_b", "grad_c", "grad_d"):
            worst["grad"] = max(worst["grad"], float(np.max(np.abs(g[k][0] - ref[k])) / (1 + np.max(np.abs(ref[k])))))
        q = rng.standard_normal((B, N))
        ys = ctx.simulate(A, Bc, C, Dd, t, s2, q)
        assert name() == "wide (step-by-step simulation)", idx
        rs = O.sim(A[B - 1], Bc[B - 1], C, Dd, t, s2, q[B - 1])
        worst["sim"] = max(worst["sim"], float(np.max(np.abs(ys[B - 1] - rs)) / np.max(np.abs(rs))))
        if True:
            tau = np.sort(np.concatenate([rng.uniform(t[0] - 5, t[-1] + 5, 40), t[[0, N - 1]]]))
            got = ds.predict(A, Bc, C, Dd, tau, mu=mu, nu=nu)
            assert name() == "wide (step-by-step prediction)", idx
            rp = O.predict(A[0], Bc[0], C, Dd, tau, t, y - mu[0], nu[0] * s2) + mu[0]
            worst["pred"] = max(worst["pred"], float(np.max(np.abs(got[0] - rp)) / max(1.0, np.max(np.abs(rp)))))
        ds.close()
    with capsys.disabled():
        print(f"\n64..143-row fuzz: 40 cases, worst deviations from the oracle {worst}", file=sys.stderr)
    assert worst["grad"] < 1e-8 and worst["pred"] < 1e-8 and worst["sim"] < 1e-9, worst


def test_fuzz_time_parallel_scan_vs_oracle(capsys):
    """Round 6: the time-parallel family with its boundary phase as a scan (forced: scan_config "tp", tp_scan 1) on 300 seeded random shapes — 2 .. 32 terms, some of
    them one-row terms (3 .. 64 state rows, padded to a multiple of 8; 49 .. 64 rows on tp_combine_lean_kernel), N = 64 .. 3000 with occasional long gaps, 1 .. 4 draws, segment
    counts 2 .. 40 or automatic, optional mu / nu — against the oracle: 1e-8 relative to max(1, |log L|) (observed ~1e-12), statuses equal; the check + repair pass is part of the
    path (a draw the scan gets wrong must come back right)."""
    import numpy as np
    import pioran_jl_amd as pj
    from oracle import oracle as O
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_gpu_parity import _random_case
    ctx = pj.Context(0)
    name = lambda: pj._lib.lib().pioran_celerite_config_name(-1).decode()
    worst, worst_case, ncases = 0.0, None, 300
    try:
        ctx.set_option("scan_config", "tp"); ctx.set_option("tp_scan", 1)
        for idx in range(ncases):
            rng = np.random.default_rng(991000 + idx)
            J = int(rng.integers(2, 33)); nreal = int(rng.integers(0, J // 2 + 1)) if rng.random() < 0.5 else 0
            N = int(rng.integers(64, 3001)); B = int(rng.integers(1, 5))
            nseg = int(rng.integers(2, 41)) if rng.random() < 0.6 else 0
            t, y, s2, A, Bc, C, Dd, mu, nu = _random_case(rng, N, J, B)
            if rng.random() < 0.3:                                   # a few long gaps
                gaps = rng.integers(1, N, size=3)
                dt = np.diff(t, prepend=t[0]); dt[gaps] *= rng.uniform(20.0, 200.0, 3); t = np.cumsum(dt)
            Dd = np.maximum(Dd, 0.05)
            Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
            use_mu = rng.random() < 0.7
            ds = pj.Dataset(t, y, s2, ctx)
            ctx.set_option("tp_segments", nseg)
            got, st = ds.logl_batch(A, Bc, C, Dd, mu=mu if use_mu else None, nu=nu if use_mu else None, return_status=True)
            assert name() == "tp", (idx, J, N, B, nreal, nseg, name())
            ref, rst = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu if use_mu else None, nu if use_mu else None, return_status=True)
            ds.close()
            assert np.array_equal(st != 0, rst != 0), (idx, st, rst)
            ok = rst == 0
            if ok.any():
                dev = float(np.max(np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))))
                if dev > worst: worst, worst_case = dev, (idx, J, N, B, nreal, nseg)
    finally:
        ctx.set_option("scan_config", None); ctx.set_option("tp_scan", -1); ctx.set_option("tp_segments", 0)
    with capsys.disabled():
        print(f"\nfuzz, time-parallel scan: {ncases} cases, worst deviation {worst:.2e} at (case, J, N, B, one-row terms, segments) = {worst_case}", file=sys.stderr)
    assert worst < 1e-8
