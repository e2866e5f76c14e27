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


class OverlappedTokenGather:
    """All-gather of every step's sampled ids that does not stall the step after it.

    Ranks never need each other's ids to continue decoding (the gathered vector feeds the
    front end), so step i's collective may still be in flight while later steps compute.
    `SLOTS` send / receive buffer pairs rotate: `send_buffer(i)` is where step i's sampler writes,
    `launch(i)` starts the asynchronous all-gather, `result(i)` / `drain()` wait for it.
    Before a pair is reused, the collective of step i - SLOTS that used it must be over: that is
    first checked from the host (an event query, no GPU work); only if it is still running does
    the compute stream wait for it.  On this hardware a wait between two streams costs the
    waiting stream several microseconds even when it is already satisfied, so the compute stream
    takes none in the steady state.  With the nccl (= RCCL) backend waiting is a stream
    dependency, not a host block.
    Equal shard sizes only (pad the local batch otherwise).  Without a process group, or on a
    single rank, the local ids are the result; `single_rank_collective=True` issues the
    collective on a one-rank group too (a rehearsal of the RCCL path on a one-GPU machine).
    """

    SLOTS = 4

    def __init__(self, n_local: int, device, dtype=torch.int32, group=None, single_rank_collective=False,
                 every: int = 1):
        """`every` > 1 batches the exchange: the ids of `every` consecutive steps travel in one
        collective, issued after the last of them (or by `flush()` / `drain()`), so its fixed cost
        (a host call, a kernel and two stream dependencies) is paid once per group of steps."""
        if every < 1:
            raise ValueError(f"every must be >= 1, got {every}")
        self.group = group
        self.n_local, self.every = n_local, every
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or single_rank_collective)
        self.world = dist.get_world_size(group) if self.active else 1
        # gloo cannot move device memory: stage through the host, synchronously (CPU tests, rehearsals)
        self.host_staged = self.active and dist.get_backend(group) == 'gloo' and torch.device(device).type == 'cuda'
        self.send = [torch.zeros((every * n_local,), dtype=dtype, device=device) for _ in range(self.SLOTS)]
        self.recv = [torch.zeros((self.world * every * n_local,), dtype=dtype, device=device) if self.active
                     else None for _ in range(self.SLOTS)]
        self.work = [None] * self.SLOTS
        self.pending = [False] * self.SLOTS      # written by send_buffer users, not yet exchanged

    def _slot(self, step: int) -> int:
        return (step // self.every) % self.SLOTS

    def send_buffer(self, step: int) -> torch.Tensor:
        slot, j = self._slot(step), step % self.every
        if j == 0 or not self.pending[slot]:
            work = self.work[slot]
            if work is not None and work.is_completed():
                self.work[slot] = None
            self._wait(slot)
        self.pending[slot] = True
        return self.send[slot][j * self.n_local:(j + 1) * self.n_local]

    def launch(self, step: int) -> None:
        """Step `step`'s ids are in its send buffer; exchanges the group when this was its last step."""
        if step % self.every == self.every - 1:
            self._exchange(self._slot(step))

    def flush(self) -> None:
        """Exchanges a group that is only partly filled (the end of a run)."""
        for slot in range(self.SLOTS):
            if self.pending[slot]:
                self._exchange(slot)

    def _exchange(self, slot: int) -> None:
        self.pending[slot] = False
        if not self.active:
            return
        if self.host_staged:
            out = torch.empty((self.recv[slot].numel(),), dtype=self.recv[slot].dtype)
            dist.all_gather_into_tensor(out, self.send[slot].cpu(), group=self.group)
            self.recv[slot].copy_(out)
            return
        self.work[slot] = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group,
                                                      async_op=True)

    def _wait(self, slot: int) -> None:
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def result(self, step: int) -> torch.Tensor:
        """Gathered ids of `step`, in rank order (valid until its buffers come round again,
        SLOTS * every steps later).  Its group must have been exchanged (`launch` of the group's
        last step, or `flush`)."""
        slot, j = self._slot(step), step % self.every
        if self.pending[slot]:
            raise RuntimeError(f"step {step}: its group of {self.every} steps has not been exchanged yet (flush())")
        self._wait(slot)
        if not self.active:
            return self.send[slot][j * self.n_local:(j + 1) * self.n_local]
        if self.every == 1:
            return self.recv[slot]
        return self.recv[slot].view(self.world, self.every, self.n_local)[:, j, :].reshape(-1)

    def drain(self) -> None:
        self.flush()
        for slot in range(self.SLOTS):
            self._wait(slot)
