#!/usr/bin/env python3
"""The dispatch ladders against the sweeps they came from (no GPU needed).
  * capi.hip tile_dispatch (pioran_tile_choice) against profiles/r05_tile_batch_sweep.txt (tools/ab_tile.py: the automatic choice with the tile
    kernel switched off against the tile kernel forced, per (model, batch size)): prints every measured line with the faster family, the library's
    choice and what the choice costs; exit status 1 if a choice is more than 5 % slower than the faster family.
usage: python tools/retune_thresholds.py [sweep file]        (re-run tools/ab_tile.py on the GPU box to refresh the sweep, then edit the ladder)"""
import ctypes, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
LINE = re.compile(r"^(\w+) N=(\d+) B=(\d+) rows=(\d+): auto \[(\w+):(\w+)\] ([\d.]+) ms .*?; tile \[tile\] ([\d.]+) ms")
PASS = {"rpl3": 4096, "rpl4": 2048, "rpl5": 1024, "rpl2": 4096 * 2, "rpl1": 4096 * 2}   # draws per pass of the step-by-step layouts at N = 1e4 (two / one / one draw(s) per wavefront ...)


def sweep_lines(path):
    for l in open(path):
        m = LINE.match(l)
        if m:
            model, N, B, rows, fam, cfg, t_auto, t_tile = m.groups()
            yield dict(model=model, N=int(N), B=int(B), rows=int(rows), family=fam, config=cfg, auto_ms=float(t_auto), tile_ms=float(t_tile))


def check(path=ROOT / "profiles" / "r05_tile_batch_sweep.txt", tol=0.05, verbose=True):
    lib = ctypes.CDLL(str(ROOT / "pioran.jl_amd" / "libpioran_hip.so"))
    lib.pioran_tile_choice.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    lib.pioran_tile_choice.restype = ctypes.c_int
    bad = []
    for r in sweep_lines(path):
        ch = lib.pioran_tile_choice(r["rows"], r["B"], 0, 0)
        if ch < 0:
            ch = lib.pioran_tile_choice(r["rows"], r["B"], PASS.get(r["config"][:4], 4096), 0)
        best = min(r["auto_ms"], r["tile_ms"])
        mine = r["tile_ms"] if ch == 1 else r["auto_ms"]
        loss = mine / best - 1.0
        if verbose:
            print(f"{r['model']:8s} rows {r['rows']:3d} B {r['B']:5d}: other {r['auto_ms']:7.2f} ms ({r['family']}), tile {r['tile_ms']:7.2f} ms -> library takes "
                  f"{'tile ' if ch == 1 else 'other'} (+{100 * loss:.1f} % over the faster one)")
        if loss > tol:
            bad.append((r, loss))
    return bad


if __name__ == "__main__":
    bad = check(Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r05_tile_batch_sweep.txt")
    for r, loss in bad:
        print(f"LADDER OFF: {r['model']} rows {r['rows']} B {r['B']}: +{100 * loss:.1f} %")
    sys.exit(1 if bad else 0)
