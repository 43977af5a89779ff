"""Batch sharding across the GPUs of one node: one process per GPU, no collective on the data path.

Every sample's chain is independent (GroupNorm, attention and convergence are per sample: SURVEY.md 8e), so a batch is
split into contiguous row blocks, each rank samples its block with its own weight replica, and the only exchange is the
final gather of (Lr0, zK, K) -- 3*H*W*4 B + 28 B per sample -- which is off the timed path.  Works with any
torch.distributed backend ("nccl" = RCCL on ROCm for GPU tensors, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def shard_rows(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n rows: the first n % world ranks get one extra row. Returns [start, stop)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_indices(n: int, world_size: int, rank: int, strided: bool = False) -> torch.Tensor:
    """Row indices owned by `rank`.  Contiguous blocks by default; `strided` deals rows round-robin (rank, rank + world, ...):
    with per-sample early exit (DRMNet.p_sample_loop) neighbouring samples tend to converge at similar steps, so dealing them
    out keeps the ranks' remaining work balanced as rows drop out."""
    if strided:
        if world_size < 1 or not (0 <= rank < world_size):
            raise ValueError("bad rank/world_size")
        return torch.arange(rank, max(n, rank), world_size)
    a, b = shard_rows(n, world_size, rank)
    return torch.arange(a, b)


def shard(t: torch.Tensor, world_size: int, rank: int, strided: bool = False) -> torch.Tensor:
    if strided:
        return t[shard_indices(t.shape[0], world_size, rank, True).to(t.device)].contiguous()
    a, b = shard_rows(t.shape[0], world_size, rank)
    return t[a:b].contiguous()


def gather_results(parts: Sequence[torch.Tensor], n_total: int, group=None, strided: bool = False) -> Optional[List[torch.Tensor]]:
    """All-gathers each per-rank result tensor (ragged along dim 0) back into full-batch order on every rank."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return list(parts)
    world = dist.get_world_size(group)
    owned = [shard_indices(n_total, world, r, strided) for r in range(world)]
    rows = max(ix.numel() for ix in owned)
    out = []
    for p in parts:
        pad = torch.zeros((rows,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
        pad[: p.shape[0]] = p
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        full = torch.empty((n_total,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
        for ix, b in zip(owned, bufs):
            full[ix.to(p.device)] = b[: ix.numel()]
        out.append(full)
    return out


def sample_sharded(sample_fn, batch_tensors: Sequence[torch.Tensor], group=None, strided: bool = False):
    """Runs ``sample_fn(*local_shards) -> tuple of tensors`` on this rank's rows and gathers the results.

    ``batch_tensors`` are full-batch inputs present on every rank.  ``strided=True`` deals rows round-robin instead of in
    contiguous blocks (use it when ``sample_fn`` exits early per sample).
    """
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    n = batch_tensors[0].shape[0]
    local = [shard(t, world, rank, strided) for t in batch_tensors]
    res = sample_fn(*local)
    if not isinstance(res, (tuple, list)):
        res = (res,)
    return gather_results(list(res), n, group, strided)
