import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pioran_jl_amd as pj
from oracle import oracle as O
N, J = 10000, 20
t, y, yerr = bench.synth_series(N)
print(open('/proc/cpuinfo').read().count('processor\t'), 'logical cpus;', os.popen("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'").read())
for nt in (1, 8, 32, 64, 128, 256):
    B = max(8 * nt, 16)
    th, f_min, f_max = bench.synth_theta(B, t, y, seed=1)
    A, Bc, C, Dd = pj.approx_batch(pj.SingleBendingPowerLaw, th[:, :3], f_min, f_max, J, th[:, 3])
    O.logl_batch(A[:nt], Bc[:nt], C, Dd, t, y, yerr ** 2, th[:nt, 5].copy(), th[:nt, 4].copy(), nthreads=nt)
    t0 = time.perf_counter(); O.logl_batch(A, Bc, C, Dd, t, y, yerr ** 2, th[:, 5].copy(), th[:, 4].copy(), nthreads=nt); dt = time.perf_counter() - t0
    print(f"threads={nt:4d} draws={B:5d} {B/dt:9.1f} evals/s  {B/dt/nt:6.2f} per thread")
