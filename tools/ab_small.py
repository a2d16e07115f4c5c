#!/usr/bin/env python3
"""A/B timing of two builds of libpioran_hip.so on ONE box for the small-batch (windowed kernel) launches: resident launches of 1, 256 and
512 draws at N = 1e4 (SHO-20 and DRWCelerite-20), alternating the libraries.  usage: python tools/ab_small.py libA.so libB.so [rounds]"""
import json, os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import torch, bench, pioran_jl_amd as pj
N = 10000
t, y, yerr = bench.synth_series(N)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
th, f_min, f_max = bench.synth_theta(512, t, y, seed=4321)
res = {}
for basis in ("SHO", "DRWCelerite"):
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, 20, th[:, 3], basis_function=basis)
    real = (Dd == 0.0) & (Bc == 0.0).all(axis=0)
    ds = pj.Dataset(t, y, yerr ** 2, ctx); ds.prepare(C, Dd, real.astype(np.int32))
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A, Bc, th[:, 5].copy(), th[:, 4].copy())]
    dout = torch.empty(512, dtype=torch.float64, device=dev)
    for B in (1, 256, 512):
        go = lambda: ds.logl_batch_dev(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, 0, dout.data_ptr(), 0)
        go(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream); go(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        res[f"{basis}_B{B}"] = float(np.median(ts))
    res[f"{basis}_sum"] = float(dout[:512].nan_to_num().sum().item())
print(json.dumps(res))
''' % str(ROOT)
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
acc = {a: [], b: []}
for r in range(rounds):
    for lib in (a, b):
        env = dict(os.environ, PIORAN_HIP_LIB=str(Path(lib).resolve()))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        acc[lib].append(json.loads(line)); print(Path(lib).name, line, flush=True)
for k in acc[a][0]:
    ma, mb = np.median([x[k] for x in acc[a]]), np.median([x[k] for x in acc[b]])
    print(f"{k}: A {ma:.4f}  B {mb:.4f}  B/A {mb / ma:.4f}")
