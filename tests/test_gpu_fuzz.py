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
