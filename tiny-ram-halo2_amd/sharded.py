"""Range-sharded MSM across the GPUs of one node.

`best_multiexp` already splits its input into contiguous chunks (one per rayon thread) and sums
the chunk results; here the chunks are one per rank: rank r owns pairs [lo_r, hi_r), runs its
local Pippenger to ONE point, the 96-byte partial points are all-gathered (RCCL over xGMI on
GPUs, gloo in the CPU tests) and every rank adds them with `trh_point_sum` (EC addition is not
a reduce op of the collective library, so gather-then-add).  No other data moves.
"""
from __future__ import annotations

import numpy as np

from . import api


def shard_range(n_total: int, rank: int, world: int):
    """contiguous split; the first (n_total % world) ranks get one extra pair"""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_points(partial: np.ndarray, device=None) -> np.ndarray:
    """partial: (12,) uint64 Jacobian point of this rank -> (world, 12) on every rank"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def sharded_msm(curve: str, local_msm, device=None) -> np.ndarray:
    """local_msm() -> (12,) partial point of this rank's range; returns the global result"""
    import torch.distributed as dist

    p = np.asarray(local_msm(), dtype=np.uint64).reshape(12)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return p
    return api.point_sum(curve, all_gather_points(p, device))


def sharded_columns(n_columns: int, local_batch, width: int = 12, device=None) -> np.ndarray:
    """Column-sharded batch step of create_proof (the commitments of the advice / permuted / product columns, or any other
    per-column result of `width` u64 words): columns are independent, so rank r takes the contiguous columns
    shard_range(n_columns, r, world), runs `local_batch(lo, hi) -> (hi - lo, width)` (e.g. Params.commit_lagrange_batch over its
    columns) and ONE all-gather of the padded blocks hands every rank all results in column order, which is what the
    transcript needs next.  96 bytes per column: latency-only over xGMI, no data-path collective."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local_batch(0, n_columns), dtype=np.uint64).reshape(n_columns, width)
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_columns, rank, world)
    mine = np.asarray(local_batch(lo, hi), dtype=np.uint64).reshape(hi - lo, width)
    per = (n_columns + world - 1) // world            # the longest shard
    block = np.zeros((per, width), dtype=np.uint64)
    block[: hi - lo] = mine
    t = torch.from_numpy(block.view(np.int64))
    if device is not None:
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = np.zeros((n_columns, width), dtype=np.uint64)
    for r in range(world):
        l, h = shard_range(n_columns, r, world)
        res[l:h] = out[r].cpu().numpy().view(np.uint64)[: h - l]
    return res
