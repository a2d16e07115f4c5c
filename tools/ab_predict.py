#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: posterior mean and simulation (tools/bench_predict.py), alternating, medians.
usage: python tools/ab_predict.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {a: [], b: []}
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, PIORAN_HIP_LIB=str(Path(lib).resolve()))
        out = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_predict.py")], env=env, capture_output=True, text=True, timeout=600)
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        res[lib].append((d["predict_ms_per_call"], d["simulate_ms_per_call"]))
        print(Path(lib).name, res[lib][-1], flush=True)
for lib in (a, b):
    m = np.median(np.array(res[lib]), axis=0)
    print(f"{Path(lib).name}: predict {m[0]:.3f} ms, simulate {m[1]:.3f} ms")
