"""CPU tests of the multi-GPU farm logic: world_size-2 (and 3, ragged) gloo process groups.
The per-rank evaluator is the oracle here (CPU stand-in for a rank's GPU): what is under test is the
sharding + gather path that bench.py --gpus N uses with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pioran_jl_amd as pj
from oracle import oracle as O


def test_shard_bounds_partition():
    for B in (0, 1, 7, 8, 4096, 32768, 32769):
        for world in (1, 2, 3, 8):
            cuts = [pj.farm.shard_bounds(B, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pj.farm.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)  # same inputs on every rank
        N, J = 80, 4
        t = np.cumsum(rng.uniform(0.1, 2.0, N)); y = rng.standard_normal(N); s2 = rng.uniform(0.01, 0.1, N)
        A = rng.uniform(0.1, 2.0, (B, J)); Bc = rng.uniform(-0.05, 0.05, (B, J)) * A
        C = rng.uniform(0.05, 2.0, J); Dd = rng.uniform(0.0, 3.0, J)
        mu = rng.standard_normal(B) * 0.1; nu = rng.uniform(0.5, 2.0, B)

        def evaluate(lo, hi):
            return O.logl_batch(A[lo:hi], Bc[lo:hi], C, Dd, t, y, s2, mu[lo:hi], nu[lo:hi])

        full = pj.farm.farm_logl(evaluate, B).numpy()
        ref = O.logl_batch(A, Bc, C, Dd, t, y, s2, mu, nu)
        ok = bool(np.array_equal(full, ref))
        if B % world == 0:   # the overlapped form bench.py uses: equal shards, preallocated output, wait on the handle
            lo, hi = pj.farm.shard_bounds(B, world, rank)
            out = torch.empty(B, dtype=torch.float64)
            pj.farm.gather_logl_async(torch.from_numpy(evaluate(lo, hi).copy()), out).wait()
            ok = ok and bool(np.array_equal(out.numpy(), ref))
        q.put((rank, ok, full.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 64), (2, 37), (3, 10), (8, 64), (8, 45)])   # 8: the driver's world size (one rank per GPU of a node)
def test_farm_gloo(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(ok for _, ok, _ in res)
    assert all(shape == (B,) for _, _, shape in res)
