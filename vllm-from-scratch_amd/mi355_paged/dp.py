"""Data-parallel sequence sharding across the GPUs of one node.

Sequences never interact on this path (each has its own block table, KV blocks
and sampler row), so a batch is split by rows: rank r owns a contiguous slice,
keeps its own KV pool, and the only exchange per decode step is an all-gather
of the sampled token ids (a few KiB, latency-bound) over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests).
The reference has no multi-device code; this layer has no counterpart there.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first (total % world_size) ranks get one extra row."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world_size: int) -> List[int]:
    return [b - a for a, b in (shard_bounds(total, world_size, r) for r in range(world_size))]


def shard_rows(items: Sequence, world_size: int, rank: int):
    a, b = shard_bounds(len(items), world_size, rank)
    return items[a:b]


def all_gather_token_ids(local_ids: torch.Tensor, total: Optional[int] = None,
                         group=None) -> torch.Tensor:
    """Concatenate every rank's sampled ids in rank order -> [total] on every rank.

    local_ids: 1-D integer tensor (this rank's slice, as produced by the sampler).
    With equal shard sizes this is one all_gather_into_tensor; ragged shards are padded
    to the largest one for the collective and trimmed afterwards.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_ids
    world = dist.get_world_size(group)
    if local_ids.is_cuda and dist.get_backend(group) == 'gloo':
        # gloo moves host memory only: stage through the CPU (CPU tests and rehearsals)
        return all_gather_token_ids(local_ids.cpu(), total, group).to(local_ids.device)
    n_local = local_ids.numel()
    if total is None:
        total = n_local * world
    sizes = shard_sizes(total, world)
    widest = max(sizes)
    send = local_ids
    if n_local != widest:
        send = torch.zeros((widest,), dtype=local_ids.dtype, device=local_ids.device)
        send[:n_local] = local_ids
    recv = torch.empty((world * widest,), dtype=local_ids.dtype, device=local_ids.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(s == widest for s in sizes):
        return recv
    return torch.cat([recv[r * widest:r * widest + s] for r, s in enumerate(sizes)])
