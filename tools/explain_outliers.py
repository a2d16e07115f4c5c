#!/usr/bin/env python3
"""Which inputs produce the largest HIP-vs-oracle deviations, and which side is closer to the truth?

  GPU box:   python tools/explain_outliers.py --gpu [seconds] --out gpurun_out/outliers.json
               (a) a long run of tools/fuzz_layouts.py keeping the worst (case, layout, draw) records;
               (b) the J = 2, B = 256 line of profiles/r02_block_sweep.txt (windowed vs throughput kernel 7.2e-10 apart where
                   every neighbour is 1e-15) regenerated with the sweep's own rng sequence.
  anywhere:  python tools/explain_outliers.py --analyze gpurun_out/outliers.json
               regenerates every recorded case from (seed, index), evaluates it in 80-bit extended precision (numpy
               longdouble, forward-only recurrence of SURVEY.md appendix A) and prints, per record, the deviation of the HIP
               value and of the fp64 oracle from that reference together with the conditioning of the draw (smallest D_n,
               cancellation in the sums).
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _fuzz_module():
    spec = importlib.util.spec_from_file_location("_fuzz_layouts", ROOT / "tools" / "fuzz_layouts.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def logl_extended(a, b, c, d, t, y, s2):
    """log L in numpy longdouble (x87 80-bit: 64-bit mantissa), forward-only form: S <- (phi phi') o (S + m m'/D), q = S u,
    D_n = sum(a) + s2_n - u'q, m = v - q, z_n = y_n - u'f (src/celerite_solver.jl:12-158 restated as in SURVEY appendix A).
    Returns (log L, min_n D_n, sum log|D|, sum z^2/D)."""
    L = np.longdouble
    a, b, c, d, t, y, s2 = (np.asarray(v, dtype=L) for v in (a, b, c, d, t, y, s2))
    J, N = len(a), len(t)
    suma = a.sum()
    S = np.zeros((2 * J, 2 * J), dtype=L); f = np.zeros(2 * J, dtype=L)
    m = np.zeros(2 * J, dtype=L); Dp = L(1); zp = L(0)
    ld = L(0); quad = L(0); dmin = np.inf
    for n in range(N):
        co, si = np.cos(d * t[n]), np.sin(d * t[n])
        u = np.empty(2 * J, dtype=L); v = np.empty(2 * J, dtype=L)
        u[0::2] = a * co + b * si; u[1::2] = a * si - b * co; v[0::2] = co; v[1::2] = si
        if n > 0:
            ph = np.repeat(np.exp(-c * (t[n] - t[n - 1])), 2)
            S = np.outer(ph, ph) * (S + np.outer(m, m) / Dp)
            f = ph * (f + m / Dp * zp)
        q = S @ u
        D = suma + s2[n] - u @ q
        m = v - q
        z = y[n] - u @ f
        ld += np.log(np.abs(D)) if n > 0 else np.log(D)
        quad += z * z / D
        dmin = min(dmin, float(D)); Dp = D; zp = z
    res = -L(0.5) * ld - L(0.5) * N * np.log(2 * np.pi * L(1)) - L(0.5) * quad
    return res, dmin, ld, quad


def gpu_part(seconds, out):
    import pioran_jl_amd as pj
    from oracle import oracle as O
    fz = _fuzz_module()
    ctx = pj.Context(0)
    rec = {"fuzz": fz.fuzz(pj, O, ctx, seed=fz.SEED, seconds=seconds, keep=12)}
    # (b) tools/sweep_block.py, J = 2: B = 1, 64, 256 drawn in this order from default_rng(3)
    N = 10_000
    t, y, yerr = O.synthetic_series(N)
    rng = np.random.default_rng(3)
    for B in (1, 64, 256):
        J = 2
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.005, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)
    ds = pj.Dataset(t, y, yerr ** 2, ctx)
    ctx.set_option("scan_config", "block"); g1 = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
    ctx.set_option("scan_config", None); ctx.set_option("no_block", True); g2 = ds.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
    ctx.set_option("no_block", False)
    ref = O.logl_batch(A, Bc, C, Dd, t, y, yerr ** 2, mu, nu, nthreads=8)
    i = int(np.argmax(np.abs(g1 - g2) / np.abs(g2)))
    rec["sweep_j2_b256"] = dict(draw=i, block=float(g1[i]), other=float(g2[i]), oracle=float(ref[i]),
                                rel_block_vs_other=float(abs(g1[i] - g2[i]) / abs(g2[i])), abs_block_vs_other=float(abs(g1[i] - g2[i])),
                                max_abs_block_vs_other_all_draws=float(np.max(np.abs(g1 - g2))),
                                max_abs_block_vs_oracle=float(np.max(np.abs(g1 - ref))), max_abs_other_vs_oracle=float(np.max(np.abs(g2 - ref))),
                                median_abs_logl=float(np.median(np.abs(ref))),
                                a=A[i].tolist(), b=Bc[i].tolist(), c=C.tolist(), d=Dd.tolist(), mu=float(mu[i]), nu=float(nu[i]))
    Path(out).parent.mkdir(parents=True, exist_ok=True)
    Path(out).write_text(json.dumps(rec, indent=1))
    print(json.dumps({k: (v if k != "fuzz" else {kk: vv for kk, vv in v.items() if kk != "worst"}) for k, v in rec.items()}))
    for r in rec["fuzz"]["worst"]:
        print("worst", r)


def analyze(path):
    from oracle import oracle as O
    fz = _fuzz_module()
    rec = json.loads(Path(path).read_text())
    print(f"fuzz: {rec['fuzz']['cases']} cases, worst deviation {rec['fuzz']['worst_dev']:.2e}")
    print("case layout  J   N   B nreal draw |   log L (ext)      | HIP - ext   oracle - ext | min D_n     sum log|D|   sum z^2/D")
    for r in rec["fuzz"]["worst"]:
        c = fz.make_case(rec["fuzz"]["seed"], r["idx"])
        ys, ss = fz.draw_series(c, r["draw"])
        ext, dmin, ld, quad = logl_extended(c["A"][r["draw"]], c["Bc"][r["draw"]], c["C"], c["Dd"], c["t"], ys, ss)
        orc = O.logl(c["A"][r["draw"]], c["Bc"][r["draw"]], c["C"], c["Dd"], c["t"], ys, ss)
        print(f"{r['idx']:5d} {r['layout']:5s} {r['J']:3d} {r['N']:4d} {r['B']:3d} {r['nreal']:3d} {r['draw']:4d} | {float(ext):18.9f} | "
              f"{float(r['got'] - ext):+.2e}  {float(orc - ext):+.2e} | {dmin:.3e} {float(ld):12.3f} {float(quad):12.3f}")
    s = rec.get("sweep_j2_b256")
    if s:
        t, y, yerr = O.synthetic_series(10_000)
        ext, dmin, ld, quad = logl_extended(s["a"], s["b"], s["c"], s["d"], t, y - s["mu"], s["nu"] * yerr ** 2)
        print(f"sweep J=2 B=256, draw {s['draw']}: log L (ext) = {float(ext):.9f}; block - ext = {float(s['block'] - ext):+.3e}, "
              f"other - ext = {float(s['other'] - ext):+.3e}, oracle - ext = {float(s['oracle'] - ext):+.3e}; "
              f"-0.5 sum log|D| = {float(-0.5 * ld):.3f}, -0.5 sum z^2/D = {float(-0.5 * quad):.3f}, min D_n = {dmin:.3e}; "
              f"median |log L| of the batch = {s['median_abs_logl']:.1f}; largest |block - other| over the batch = "
              f"{s['max_abs_block_vs_other_all_draws']:.3e}")


if __name__ == "__main__":
    if "--gpu" in sys.argv:
        i = sys.argv.index("--gpu")
        secs = float(sys.argv[i + 1]) if i + 1 < len(sys.argv) and not sys.argv[i + 1].startswith("--") else 120.0
        out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else str(ROOT / "gpurun_out" / "outliers.json")
        gpu_part(secs, out)
    elif "--analyze" in sys.argv:
        analyze(sys.argv[sys.argv.index("--analyze") + 1])
    else:
        print(__doc__)
