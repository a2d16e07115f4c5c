#!/usr/bin/env python3
"""The reference's own benchmark grid (benchmark/benchmarks.jl:16-18, 74-91: group "celerite_likelihood") on one MI355X:
j in 2^(1..6) celerite terms, N in 2^(5..16) time stamps, random (a, b, c, d) = rand(j) with a *= 5, `logl(a, b, c, d, t[1:N],
y[1:N], yerr[1:N])` (the suite passes yerr, not yerr^2, as the variance: kept).  The reference times ONE scalar call; here
  (i)   scalar_ms      the scalar drop-in pioran_celerite_logl (host vectors in, one double out: PCIe included), median
  (ii)  perdraw_rate   B draws with PER-DRAW (c, d) (each draw = one of the reference's calls with its own random
                       coefficients) through the device-pointer entry: evaluations per second
  (iii) shared_rate    B draws sharing (c, d) (the approx / sampler situation: one table): evaluations per second
  (iv)  cpu_ms         the oracle (reference algorithm and memory layout) on ONE host core, the reference's setting
Series: the synthetic irregular series of bench.py at N = 65536 (benchmark/simulate_long.txt is not in the checkout).
j = 64 (128 rows) is past the register-resident kernels (95 rows): it runs on the any-rank kernel with S in HBM, with a
smaller batch.  Prints one JSON document; profiles/r02_grid.json is a committed run."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, pioran_jl_amd as pj
from oracle import oracle as O

NS = [2 ** k for k in range(5, 17)]
JS = [2 ** k for k in range(1, 7)]
B_BIG = int(os.environ.get("B", 4096))
t, y, yerr = bench.synth_series(NS[-1])
rng = np.random.Generator(np.random.PCG64(1234))                      # benchmarks.jl:27 seeds MersenneTwister(1234)
abcd = rng.random((JS[-1], 4)); abcd[:, 0] *= 5                       # :74-76
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
ctx = pj.Context(0, stream=stream.cuda_stream)
L = pj._lib.lib()
v = ctypes.c_void_p


def med_ms(f, reps):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


rows = []
for j in JS:
    a, b, c, d = (np.ascontiguousarray(abcd[:j, k]) for k in range(4))
    B = B_BIG if j <= 32 else 256
    # per-draw coefficient sets: the reference's vector for draw 0, fresh rand(j, 4) rows for the others (same distribution)
    ABCD = rng.random((B, j, 4)); ABCD[:, :, 0] *= 5; ABCD[0] = abcd[:j]
    dA, dB_, dC, dD = (torch.from_numpy(np.ascontiguousarray(ABCD[:, :, k])).to(dev) for k in range(4))
    dout = torch.empty(B, dtype=torch.float64, device=dev); dst = torch.zeros(B, dtype=torch.int32, device=dev)
    for N in NS:
        if j == 64 and N > 8192:
            continue
        tt, yy, ss = t[:N], y[:N], yerr[:N]                           # yerr as the variance argument, as benchmarks.jl:89
        ds = pj.Dataset(tt, yy, ss, ctx)
        reps = 5 if N <= 8192 else 3
        scalar = med_ms(lambda: ctx.logl(a, b, c, d, tt, yy, ss), reps)
        val = ctx.logl(a, b, c, d, tt, yy, ss)
        t0 = time.perf_counter(); ref = O.logl(a, b, c, d, tt, yy, ss); cpu = (time.perf_counter() - t0) * 1e3
        if N <= 2048:
            t0 = time.perf_counter()
            for _ in range(5): O.logl(a, b, c, d, tt, yy, ss)
            cpu = (time.perf_counter() - t0) * 1e3 / 5
        def perdraw():
            pj._lib.check(L.pioran_celerite_logl_batch_dev_cd(ds._h, B, j, v(dA.data_ptr()), v(dB_.data_ptr()), v(dC.data_ptr()),
                                                             v(dD.data_ptr()), None, None, None, None, v(dout.data_ptr()), v(dst.data_ptr())), ctx._h)
        pd = med_ms(perdraw, 2 if N >= 16384 else 3)
        first = float(dout[0].item())
        ds.prepare(c, d)
        def shared():
            ds.logl_batch_dev(B, dA.data_ptr(), dB_.data_ptr(), 0, 0, 0, 0, dout.data_ptr(), dst.data_ptr())
        sh = med_ms(shared, 2 if N >= 16384 else 3)
        err = abs(val - ref) / abs(ref) if np.isfinite(ref) and np.isfinite(val) else None
        rows.append({"j": j, "N": N, "scalar_ms": round(scalar, 4), "cpu_ms": round(cpu, 4), "B": B,
                     "perdraw_evals_per_s": round(B / pd * 1e3, 1), "shared_evals_per_s": round(B / sh * 1e3, 1),
                     "rel_err_scalar_vs_oracle": err, "perdraw_first_equals_scalar": bool(first == val) or (abs(first - val) <= 1e-9 * abs(val))})
        ds.close()
        print(rows[-1], file=sys.stderr, flush=True)

# the values read off the reference's figure (+-15 %, unstated CPU, one thread)
published = {"(2, 8192)": 0.85, "(4, 8192)": 1.6, "(8, 8192)": 3.7, "(16, 8192)": 10.0, "(32, 8192)": 32.0, "(64, 8192)": 180.0, "(16, 65536)": 80.0}
print(json.dumps({"workload": "benchmark/benchmarks.jl celerite_likelihood grid; synthetic series; random (a, b, c, d)", "rows": rows,
                  "reference_figure_ms": published, "host_cpu_count": os.cpu_count()}))
