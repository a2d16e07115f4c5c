"""GPU test (-m gpu) of the process-per-GPU farm on REAL devices: `python bench.py --gpus N` — the exact command the driver
runs, NO launcher in the test: bench.py starts its own ranks (bench.spawn_ranks: torch.distributed.run as a child process)
— with one rank per GPU over RCCL: BASELINE.json configs[3] (N = 1e4, J = 20, 4096 draws per GPU, all-gather of log L).
The RCCL test is skipped on boxes with fewer than two GPUs; tests/test_farm.py covers the same sharding / gather logic with
gloo on CPU, tests/test_bench_spawn.py the launcher-less start on CPU, and
tests/test_gpu_configs.py::test_config4_global_batch_on_one_gpu the 8-way cut at full size on one device.

bench.py runs as a fresh CHILD process (subprocess): the pytest process itself is never replaced.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
NGPU = torch.cuda.device_count()          # counting devices does not initialise the GPU


def _bench(n, extra=(), steps=3):
    """`python bench.py --gpus n ...` exactly as the driver types it: no torch.distributed.run here."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline",
           "--no-secondary", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_multi_rank_loop_two_ranks_on_one_gpu():
    """bench.py's multi-rank loop (double-buffered scan / gather, slice check, --verify-gather, MAX-reduced timing) run for real
    on ONE GPU: `python bench.py --gpus 2` starts its own two ranks, both on cuda:0, log-L gathered over gloo (RCCL refuses
    two ranks on one device; the transport is the only difference from the 8-GPU run)."""
    one = _bench(1, ["--batch", "1024"], steps=4)
    # the spawned path at N = 1 (one rank under the child launcher, collective path taken) against the plain process
    one_spawned = _bench(1, ["--batch", "1024", "--spawn", "--dist-backend", "gloo"], steps=4)
    assert one_spawned["n_gpus"] == 1 and 0.8 * one["value"] < one_spawned["value"] < 1.25 * one["value"], (one["value"], one_spawned["value"])
    two = _bench(2, ["--dist-backend", "gloo", "--device", "0", "--batch", "1024", "--verify-gather"], steps=4)
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 2048 and two["steps"] == 4
    assert two["gather_verified"] is True          # rank 1's slice == a single-process evaluation of rank 1's batch, bit for bit
    assert two["status_ok_frac"] > 0.9
    assert "gloo" in two["config"]["parallelism"]
    # both ranks share one GPU, so whole-job throughput is about the one-rank figure (not twice it); the timed region must
    # cover both ranks' work: value = global draws / MAX-over-ranks time
    assert 0.4 * one["value"] < two["value"] < 2.5 * one["value"], (one["value"], two["value"])
    assert two["ms_per_step"] >= 0.8 * one["ms_per_step"]


def test_bench_multi_rank_loop_three_ranks_on_one_gpu():
    """The same loop at an odd world size: three ranks on cuda:0 over gloo.  (One GPU of this pool admits six processes on its card at once,
    and this pytest process, bench.py's parent and the launcher count: five ranks were refused by the box's process guard.  The driver's eight
    ranks need eight GPUs — their rendezvous, thread division and gather order at world size 8 are exercised on the CPU:
    tests/test_bench_spawn.py::test_spawn_ranks_at_the_drivers_world_size, tests/test_farm.py.)  Per-rank seeds, the rank-ordered gather
    and the MAX-reduced time with ranks that finish apart."""
    three = _bench(3, ["--dist-backend", "gloo", "--device", "0", "--batch", "512", "--verify-gather"], steps=3)
    assert three["n_gpus"] == 3 and three["config"]["global_batch"] == 1536 and three["steps"] == 3
    assert three["gather_verified"] is True
    assert three["status_ok_frac"] > 0.9


@pytest.mark.skipif(NGPU < 2, reason="needs at least two GPUs (RCCL ranks)")
def test_bench_over_rccl_ranks():
    n = min(8, NGPU)
    one = _bench(1)
    many = _bench(n, ["--verify-gather"])
    assert many["n_gpus"] == n and many["config"]["global_batch"] == n * one["config"]["batch_per_gpu"]
    # rank 0 re-evaluated every rank's batch on its own GPU: the gathered slices are the single-GPU values, bit for bit
    assert many["gather_verified"] is True
    assert many["status_ok_frac"] > 0.9
    # weak scaling: the only exchange is a 32 KiB-per-rank all-gather overlapped with the next scan
    assert many["value"] >= 0.75 * n * one["value"], (many["value"], one["value"])


@pytest.mark.skipif(NGPU < 2, reason="needs at least two GPUs")
def test_in_process_farm_over_real_devices():
    """pioran_farm_* (one host thread + context per device inside ONE process) across distinct GPUs."""
    import numpy as np

    import pioran_jl_amd as pj
    from oracle import oracle as O
    t, y, yerr = O.synthetic_series(2000, seed=5)
    B = 1000 * NGPU + 3
    th = O.synthetic_theta(B, t, y, seed=6)
    A, Bc, C, Dd, mu, nu = O.theta_to_coefs(th, t, 20, "SHO")
    farm = pj.Farm(list(range(NGPU)), t, y, yerr ** 2)
    got, st = farm.logl_batch(A, Bc, C, Dd, mu=mu, nu=nu, return_status=True)
    farm.close()
    one = pj.Dataset(t, y, yerr ** 2, pj.Context(0)).logl_batch(A, Bc, C, Dd, mu=mu, nu=nu)
    ok = st == 0
    assert ok.mean() > 0.9 and np.array_equal(got[ok], one[ok])
