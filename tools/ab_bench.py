#!/usr/bin/env python3
"""A/B timing of two builds of libpioran_hip.so ON THE SAME GPU BOX (clock and device differences between boxes are several per
cent — larger than most kernel changes): runs bench.py alternately with PIORAN_HIP_LIB = A and B, `rounds` times each, and
prints kernel_ms per run and the medians.  usage: python tools/ab_bench.py libA.so libB.so [rounds] [bench args ...]"""
import json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {a: [], b: []}
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, PIORAN_HIP_LIB=str(Path(lib).resolve()))
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--no-secondary", "--steps", "10", "--warmup", "3", *extra],
                             env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        res[lib].append(json.loads(line)["roofline"]["kernel_ms"])
        print(Path(lib).name, f"{res[lib][-1]:.3f} ms", flush=True)
ma, mb = np.median(res[a]), np.median(res[b])
print(f"A {Path(a).name}: median {ma:.3f} ms; B {Path(b).name}: median {mb:.3f} ms; B/A = {mb / ma:.4f}")
