#!/usr/bin/env python3
"""GPU: randomized comparison of the windowed kernel and the throughput scan (two-step form) against the CPU oracle — random
term counts (1..31, some of them one-row terms), series lengths 1..700 with occasional long gaps, batch sizes, optional mu / nu /
per-draw series.  Every case is generated from (seed, index) alone, so the worst ones can be regenerated anywhere (also on a
box without a GPU, for an extended-precision look at them: tools/explain_outliers.py).

usage: python tools/fuzz_layouts.py [seconds] [--dump gpurun_out/fuzz_worst.json] [--terms LO HI]
  --terms 32 47: 64 .. 94 rows less the one-row terms (windowed kernel with five / six block columns, round 4; the throughput
  scan's 80- and 96-row shapes)
  round 2: 26 533 cases in 150 s, worst relative deviation 2.6e-9, no status / NaN mismatch
tests/test_gpu_fuzz.py runs a case-bounded slice of the same generator under -m gpu."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

SEED = 20261003
LAYOUTS = (("block", "block"), ("scan", None))


def make_case(seed: int, idx: int, jrange=(1, 32)) -> dict:
    """jrange = (lowest, highest + 1) number of terms; the default reproduces the cases of rounds 2 and 3 bit for bit."""
    rng = np.random.default_rng([seed, idx])
    J = int(rng.integers(*jrange)); N = int(rng.integers(1, 700)); B = int(rng.integers(1, 40))
    nreal = int(rng.integers(0, J + 1)) if rng.random() < 0.4 else 0
    t = np.cumsum(rng.uniform(0.01, 3.0, N) * (rng.random(N) < 0.9) + rng.uniform(0, 40, N) * (rng.random(N) < 0.05) + 1e-3)
    y = rng.standard_normal(N); s2 = rng.uniform(1e-4, 0.1, N)
    A = rng.uniform(0.05, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
    C = np.exp(rng.uniform(np.log(1e-3), np.log(50.0), J)); Dd = rng.uniform(0.0, 20.0, J)
    Bc[:, :nreal] = 0.0; Dd[:nreal] = 0.0
    mu = rng.standard_normal(B) * 0.1 if rng.random() < 0.7 else None
    nu = rng.uniform(0.5, 2.0, B) if rng.random() < 0.7 else None
    useY = bool(rng.random() < 0.3)
    Y = rng.standard_normal((B, N)) if useY else None
    S2 = rng.uniform(1e-4, 0.1, (B, N)) if useY else None
    return dict(idx=idx, J=J, N=N, B=B, nreal=nreal, useY=useY, t=t, y=y, s2=s2, A=A, Bc=Bc, C=C, Dd=Dd, mu=mu, nu=nu, Y=Y, S2=S2)


def draw_series(c: dict, i: int):
    """(y - mu, nu * sigma2) of draw i: what the reference's logl is called with (src/scalable_GP.jl:163-165)."""
    m = c["mu"][i] if c["mu"] is not None else 0.0
    v = c["nu"][i] if c["nu"] is not None else 1.0
    return (c["Y"][i] if c["useY"] else c["y"]) - m, v * (c["S2"][i] if c["useY"] else c["s2"])


def oracle_values(O, c: dict) -> np.ndarray:
    out = np.empty(c["B"])
    for i in range(c["B"]):
        ys, ss = draw_series(c, i)
        out[i] = O.logl(c["A"][i], c["Bc"][i], c["C"], c["Dd"], c["t"], ys, ss)
    return out


def gpu_values(pj, ctx, c: dict) -> dict:
    ds = pj.Dataset(c["t"], c["y"], c["s2"], ctx)
    res = {}
    try:
        for name, cfg in LAYOUTS:
            ctx.set_option("scan_config", cfg); ctx.set_option("no_block", cfg is None); ctx.set_option("no_wide", cfg is None)
            res[name] = ds.logl_batch(c["A"], c["Bc"], c["C"], c["Dd"], mu=c["mu"], nu=c["nu"], Y=c["Y"], S2=c["S2"], return_status=True)
    finally:
        ctx.set_option("scan_config", None); ctx.set_option("no_block", False); ctx.set_option("no_wide", False)
    return res


def fuzz(pj, O, ctx, seed=SEED, ncases=None, seconds=None, first=0, keep=8, jrange=(1, 32)):
    """Runs cases first, first + 1, ... until `ncases` are done or `seconds` have passed.  Returns a summary: the number of
    cases, the worst deviation relative to max(1, |log L|), the `keep` worst (case, layout, draw) records, the failures."""
    t0 = time.time(); it = 0; worst = []; failures = []
    while (ncases is None or it < ncases) and (seconds is None or time.time() - t0 < seconds):
        c = make_case(seed, first + it, jrange); it += 1
        ref = oracle_values(O, c)
        for name, (got, st) in gpu_values(pj, ctx, c).items():
            ok = np.isfinite(ref) & (st == 0)
            if (np.isnan(got) != np.isnan(ref)).any():
                failures.append(dict(kind="nan-mismatch", layout=name, idx=c["idx"], J=c["J"], N=c["N"], B=c["B"], nreal=c["nreal"]))
            if not ok.any():
                continue
            dev = np.where(ok, np.abs(got - ref) / np.maximum(1.0, np.abs(ref)), 0.0)
            i = int(np.argmax(dev))
            rec = dict(dev=float(dev[i]), layout=name, idx=c["idx"], draw=i, J=c["J"], N=c["N"], B=c["B"], nreal=c["nreal"], useY=c["useY"],
                       got=float(got[i]), oracle=float(ref[i]))
            worst.append(rec); worst.sort(key=lambda r: -r["dev"]); del worst[keep:]
            if rec["dev"] > 1e-8:
                failures.append(dict(kind="deviation", **rec))
    return dict(seed=seed, first=first, cases=it, seconds=time.time() - t0, worst_dev=worst[0]["dev"] if worst else 0.0, worst=worst,
                failures=failures)


if __name__ == "__main__":
    import pioran_jl_amd as pj
    from oracle import oracle as O
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if args else 150.0
    jr = (1, 32)
    if "--terms" in sys.argv:
        k = sys.argv.index("--terms"); jr = (int(sys.argv[k + 1]), int(sys.argv[k + 2]) + 1); args = [a for a in args if a not in sys.argv[k + 1:k + 3]]
        budget = float(args[0]) if args else 150.0
    s = fuzz(pj, O, pj.Context(0), seconds=budget, jrange=jr)
    for r in s["worst"]:
        print("worst", r)
    for f in s["failures"]:
        print("BAD", f, flush=True)
    print("cases", s["cases"], "worst relative deviation", s["worst_dev"], "failures", len(s["failures"]))
    if "--dump" in sys.argv:
        Path(sys.argv[sys.argv.index("--dump") + 1]).write_text(json.dumps(s, indent=1))
    sys.exit(1 if s["failures"] else 0)
