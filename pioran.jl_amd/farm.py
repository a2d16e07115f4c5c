"""Batch farm over the GPUs of one node: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The unit of work is one independent log-likelihood evaluation (a nested-sampling live point / MCMC
walker) — the reference farms these over MPI ranks or Distributed workers, one scalar logpdf per call
(docs/src/ultranest.md:143-149, docs/src/turing.md:98-130).  Here rank g evaluates the contiguous
slice [lo_g, hi_g) of the B draws on its own GPU with no data-path communication; the only collective
is the final all-gather of the B log-L values (<= 32 KiB per rank at B = 32768: latency-, not
bandwidth-bound, so xGMI link bandwidth is irrelevant here).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of B draws: the first B % world ranks get one extra draw."""
    if B < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_logl(local, B: int, group=None, device=None):
    """All-gather the per-rank slices (possibly ragged, sizes from shard_bounds) into the full (B,) vector.
    `local` is a torch tensor (on the GPU for nccl, CPU for gloo) or a numpy array."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.asarray(local, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    t = t.to(torch.float64).contiguous()
    lo, hi = shard_bounds(B, world, rank)
    if t.numel() != hi - lo:
        raise ValueError(f"rank {rank}: expected {hi - lo} local values, got {t.numel()}")
    width = -(-B // world) if B else 0
    if B % world == 0:
        out = torch.empty(B, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out
    # ragged: pad every slice to the widest one, gather, then compact
    pad = torch.full((width,), float("nan"), dtype=torch.float64, device=t.device)
    pad[: t.numel()] = t
    buf = torch.empty(world * width, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for g in range(world):
        glo, ghi = shard_bounds(B, world, g)
        parts.append(buf[g * width: g * width + (ghi - glo)])
    return torch.cat(parts)


def gather_logl_async(local, out, group=None):
    """Non-blocking form for equal shards: starts the all-gather of the torch tensor `local` into the preallocated
    `out` (world * local.numel() values) on the collective's own stream and returns the work handle; `work.wait()` makes
    the CURRENT stream wait for it.  Lets the gather of batch k overlap the scan of batch k + 1 (both buffers must stay
    untouched until then: double-buffer them)."""
    import torch.distributed as dist

    if out.numel() != local.numel() * dist.get_world_size(group):
        raise ValueError("out must hold world * local.numel() values")
    return dist.all_gather_into_tensor(out, local, group=group, async_op=True)


def farm_logl(evaluate: Callable[[int, int], "np.ndarray"], B: int, group=None, device=None):
    """Evaluate draws [lo, hi) of this rank with `evaluate(lo, hi)` (e.g. a Dataset.logl_batch closure on
    this rank's GPU) and return the full (B,) log-L vector on every rank."""
    import torch.distributed as dist

    lo, hi = shard_bounds(B, dist.get_world_size(group), dist.get_rank(group))
    local = evaluate(lo, hi)
    return gather_logl(local, B, group=group, device=device)
