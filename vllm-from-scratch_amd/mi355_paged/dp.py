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


class NativeComm:
    """The library's own RCCL communicator (mi355_dp_*): one rank per process, bound to the
    current HIP device.  A torch.distributed group (any backend) is used once, to hand rank 0's
    unique id to the other ranks; the collectives themselves are single library calls on a HIP
    stream, with none of torch.distributed's per-call stream switching.
    """

    def __init__(self, device, group=None):
        import ctypes
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self.device = torch.device(device)
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            self._check(self._lib.mi355_dp_unique_id(uid), 'mi355_dp_unique_id')
        if self.world > 1:
            box = [uid.raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self._lib.mi355_dp_init(uid, self.world, self.rank, ctypes.byref(comm)), 'mi355_dp_init')
        self._comm = comm

    def all_gather_i32(self, send: torch.Tensor, recv: torch.Tensor, stream=None) -> None:
        """recv[r * n : (r + 1) * n] = rank r's send[:n]; asynchronous on `stream` (default: current)."""
        if send.dtype != torch.int32 or recv.dtype != torch.int32 or not send.is_cuda or not recv.is_cuda \
                or not send.is_contiguous() or not recv.is_contiguous() or recv.numel() != send.numel() * self.world:
            raise ValueError("send / recv must be contiguous int32 device vectors, recv world times as long")
        stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.mi355_dp_allgather_i32(self._comm, send.data_ptr(), recv.data_ptr(),
                                                         send.numel(), stream.cuda_stream), 'mi355_dp_allgather_i32')

    def device_started_exchange(self) -> "DeviceStartedExchange":
        """An exchange the GPU starts by itself once a step's ids are written (mi355_dp_exchange_*)."""
        return DeviceStartedExchange(self)

    def close(self) -> None:
        if self._comm is not None and self._comm.value:
            self._check(self._lib.mi355_dp_destroy(self._comm), 'mi355_dp_destroy')
        self._comm = None


class DeviceStartedExchange:
    """mi355_dp_exchange_*: the all-gather runs on a side stream the library owns, behind a one-wave
    watcher kernel that returns once every id of the step has been written; afterwards the send
    buffer is marked "not written" (-1) again.  The stream that samples carries nothing for the
    exchange -- no event record, no wait, no extra launch -- and its sampler is the plain one.
    Ids must be non-negative."""

    SLOTS = 8

    def __init__(self, comm: NativeComm):
        import ctypes
        self.comm, self._lib, self._check = comm, comm._lib, comm._check
        handle = ctypes.c_void_p()
        with torch.cuda.device(comm.device):
            self._check(self._lib.mi355_dp_exchange_create(comm._comm, ctypes.byref(handle)), 'mi355_dp_exchange_create')
        self._x = handle

    def _vectors(self, send, recv=None):
        ok = send.dtype == torch.int32 and send.is_cuda and send.is_contiguous()
        if recv is not None:
            ok = ok and recv.dtype == torch.int32 and recv.is_cuda and recv.is_contiguous() \
                and recv.numel() == send.numel() * self.comm.world
        if not ok:
            raise ValueError("send / recv must be contiguous int32 device vectors, recv world times as long")

    def arm(self, send: torch.Tensor) -> None:
        """Marks a send buffer as not written, once, before the first launch that reads it (synchronous)."""
        self._vectors(send)
        with torch.cuda.device(self.comm.device):
            self._check(self._lib.mi355_dp_exchange_arm(self._x, send.data_ptr(), send.numel()), 'mi355_dp_exchange_arm')

    def launch(self, send: torch.Tensor, recv: torch.Tensor, slot: int) -> None:
        """All-gather `send` into `recv` as soon as every entry of `send` has been written."""
        self._vectors(send, recv)
        with torch.cuda.device(self.comm.device):
            self._check(self._lib.mi355_dp_exchange_launch(self._x, send.data_ptr(), recv.data_ptr(), send.numel(), slot),
                        'mi355_dp_exchange_launch')

    def launch_after(self, send: torch.Tensor, recv: torch.Tensor, slot: int, stream=None) -> None:
        """All-gather `send` into `recv` once `stream` (default: current) has reached this point: one
        event record without a system fence on that stream, everything else on the side stream."""
        self._vectors(send, recv)
        stream = stream if stream is not None else torch.cuda.current_stream(self.comm.device)
        with torch.cuda.device(self.comm.device):
            self._check(self._lib.mi355_dp_exchange_launch_after(self._x, stream.cuda_stream, send.data_ptr(),
                                                                 recv.data_ptr(), send.numel(), slot),
                        'mi355_dp_exchange_launch_after')

    def done(self, slot: int) -> bool:
        rc = self._lib.mi355_dp_exchange_done(self._x, slot)
        if rc < 0:
            self._check(rc, 'mi355_dp_exchange_done')
        return rc == 1

    def wait(self, slot: int, host: bool = False) -> None:
        stream = torch.cuda.current_stream(self.comm.device).cuda_stream
        with torch.cuda.device(self.comm.device):
            self._check(self._lib.mi355_dp_exchange_wait(self._x, slot, stream, 1 if host else 0), 'mi355_dp_exchange_wait')

    def close(self) -> None:
        if self._x is not None and self._x.value:
            self._check(self._lib.mi355_dp_exchange_destroy(self._x), 'mi355_dp_exchange_destroy')
        self._x = None


class _SlotWork:
    """A DeviceStartedExchange slot as the collective handle OverlappedTokenGather keeps per buffer pair."""

    def __init__(self, exchange: DeviceStartedExchange, slot: int):
        self.exchange, self.slot = exchange, slot

    def is_completed(self) -> bool:
        return self.exchange.done(self.slot)

    def wait(self) -> None:
        self.exchange.wait(self.slot)

    def host_wait(self) -> None:
        self.exchange.wait(self.slot, host=True)


class _EventWork:
    """What OverlappedTokenGather needs of a collective handle, for a native collective issued on
    a side stream: has it finished (host-side query), and make the current stream wait for it."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def is_completed(self) -> bool:
        return self.event.query()

    def wait(self) -> None:
        torch.cuda.current_stream(self.device).wait_event(self.event)

    def host_wait(self) -> None:
        self.event.synchronize()


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
                 every: int = 1, native: Optional[NativeComm] = None, device_started: bool = True):
        """`every` > 1 batches the exchange: the ids of `every` consecutive steps travel in one
        collective, issued after the last of them (or by `flush()` / `drain()`), so its fixed cost
        (a host call, a kernel and two stream dependencies) is paid once per group of steps.
        `native`: a NativeComm to run the collective through, on a side stream of the library's
        (mi355_dp_exchange_*), instead of torch.distributed.  With `device_started` (the default;
        one step per collective, ids non-negative) the GPU starts each collective itself once the
        step's ids are written and the compute stream carries nothing for the exchange;
        `device_started=False` puts one event record (without a system fence) per exchange on the
        compute stream instead."""
        if every < 1:
            raise ValueError(f"every must be >= 1, got {every}")
        self.group = group
        self.n_local, self.every = n_local, every
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or single_rank_collective)
        self.world = dist.get_world_size(group) if self.active else 1
        self.native = native
        self.side = None
        self.device_started: Optional[DeviceStartedExchange] = None
        if native is not None:
            if dtype != torch.int32:
                raise ValueError("the native collective moves int32 ids")
            self.active, self.world = native.world > 1 or single_rank_collective, native.world
            if self.active:
                self.device_started = native.device_started_exchange()
            # (created after the library's side stream: with few hardware queues per process the
            # streams made first are the ones that get a queue of their own)
            self.side = torch.cuda.Stream(device=device)
        # gloo cannot move device memory: stage through the host, synchronously (CPU tests, rehearsals)
        self.host_staged = (self.active and native is None and dist.get_backend(group) == 'gloo'
                            and torch.device(device).type == 'cuda')
        if self.active and native is None and torch.device(device).type == 'cuda' and not self.host_staged:
            self.side = torch.cuda.Stream(device=device)
        self.send = [torch.zeros((every * n_local,), dtype=dtype, device=device) for _ in range(self.SLOTS)]
        self.recv = [torch.zeros((self.world * every * n_local,), dtype=dtype, device=device) if self.active
                     else None for _ in range(self.SLOTS)]
        self.work = [None] * self.SLOTS
        self.pending = [False] * self.SLOTS      # written by send_buffer users, not yet exchanged
        self._host = None                        # host_buffer(): one staging vector per buffer pair
        self._host_copied = [None] * self.SLOTS  # event behind the last upload of each staging vector
        self._host_graphs = None                 # launch_from_host(): captured upload + collective per pair
        self.watcher = device_started and every == 1      # else: an event record per exchange
        if self.device_started is not None and self.watcher:
            for send in self.send:
                self.device_started.arm(send)

    def _slot(self, step: int) -> int:
        return (step // self.every) % self.SLOTS

    def send_buffer(self, step: int) -> torch.Tensor:
        slot, j = self._slot(step), step % self.every
        if j == 0 or not self.pending[slot]:
            work = self.work[slot]
            if work is not None and not work.is_completed() and hasattr(work, 'host_wait'):
                # The host has run SLOTS exchanges ahead of the device.  It waits here itself -- the
                # device still has the steps in between queued -- instead of putting a wait on the
                # compute stream, which would cost that stream several microseconds on every step
                # from then on (a host that issues steps faster than the device runs them stays ahead).
                work.host_wait()
            if work is not None and work.is_completed():
                self.work[slot] = None
            self._wait(slot)
        self.pending[slot] = True
        return self.send[slot][j * self.n_local:(j + 1) * self.n_local]

    def launch(self, step: int) -> None:
        """Step `step`'s ids are in its send buffer; exchanges the group when this was its last step."""
        if step % self.every == self.every - 1:
            self._exchange(self._slot(step))

    def host_buffer(self, step: int) -> torch.Tensor:
        """Host-side staging vector of step `step`'s buffer pair (pinned when the ids travel from
        device memory), for front ends whose ids are host values: fill it, then `launch_from_host`.
        It may be rewritten once its last upload is over, which this call waits for."""
        if self.every != 1:
            raise ValueError("host staging serves one step per collective")
        slot = self._slot(step)
        if self._host is None:
            pin = self.send[0].is_cuda
            self._host = [torch.zeros((self.n_local,), dtype=self.send[0].dtype, pin_memory=pin)
                          for _ in range(self.SLOTS)]
        copied = self._host_copied[slot]
        if copied is not None:
            copied.synchronize()
            self._host_copied[slot] = None
        return self._host[slot]

    def launch_from_host(self, step: int) -> None:
        """Uploads `host_buffer(step)` and exchanges it.  With the library's communicator the upload
        and the collective are one HIP graph per buffer pair, replayed on the side stream: the
        compute stream is not involved at all and the host makes one graph launch."""
        slot = self._slot(step)
        host = self._host[slot]
        if self.native is None or not self.active:
            self.send_buffer(step).copy_(host, non_blocking=True)
            if host.is_pinned():
                self._host_copied[slot] = torch.cuda.Event()
                self._host_copied[slot].record()
            self.launch(step)
            return
        dev = self.native.device
        if self._host_graphs is None:
            self._host_graphs = self._capture_host_exchanges()
        with torch.cuda.stream(self.side):
            if self._host_graphs:
                self._host_graphs[slot].replay()
            else:
                self.send[slot].copy_(host, non_blocking=True)
                self.native.all_gather_i32(self.send[slot], self.recv[slot], stream=self.side)
            done = torch.cuda.Event()
            done.record(self.side)
        self.work[slot] = _EventWork(done, dev)
        self._host_copied[slot] = done

    def _capture_host_exchanges(self):
        """One graph per buffer pair: pinned host vector -> send buffer -> all-gather.  An empty
        list (eager calls instead) when the capture is refused."""
        graphs = []
        try:
            with torch.cuda.stream(self.side):
                for slot in range(self.SLOTS):       # first use of the communicator outside a capture
                    self.native.all_gather_i32(self.send[slot], self.recv[slot], stream=self.side)
                self.side.synchronize()
                for slot in range(self.SLOTS):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.side, capture_error_mode='thread_local'):
                        self.send[slot].copy_(self._host[slot], non_blocking=True)
                        self.native.all_gather_i32(self.send[slot], self.recv[slot],
                                                   stream=torch.cuda.current_stream(self.native.device))
                    graphs.append(g)
        except Exception:
            torch.cuda.synchronize(self.native.device)
            return []
        return graphs

    def flush(self) -> None:
        """Exchanges a group that is only partly filled (the end of a run)."""
        for slot in range(self.SLOTS):
            if self.pending[slot]:
                self._exchange(slot)

    def _exchange(self, slot: int) -> None:
        self.pending[slot] = False
        if not self.active:
            return
        if self.device_started is not None:
            # the library's side stream watches the ids arrive, then gathers and marks the buffer again
            if self.watcher:
                self.device_started.launch(self.send[slot], self.recv[slot], slot)
            else:
                self.device_started.launch_after(self.send[slot], self.recv[slot], slot)
            self.work[slot] = _SlotWork(self.device_started, slot)
            return
        if self.host_staged:
            out = torch.empty((self.recv[slot].numel(),), dtype=self.recv[slot].dtype)
            dist.all_gather_into_tensor(out, self.send[slot].cpu(), group=self.group)
            self.recv[slot].copy_(out)
            return
        if self.side is None:
            self.work[slot] = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group,
                                                          async_op=True)
            return
        # RCCL through torch.distributed, called from a side stream: the process group orders its own
        # stream behind the stream it is CALLED on, so the compute stream carries one event record
        # per exchange and none of the group's bookkeeping (one-rank rehearsal: 200 -> 195 us per step;
        # what remains is the host's 110 us per call, most of it under the interpreter lock -- issuing
        # the call from a thread of its own was measured and did not help)
        ready = torch.cuda.Event()
        ready.record()
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True).wait()
            done = torch.cuda.Event()       # (wait() above orders the side stream behind the collective)
            done.record(self.side)
        self.work[slot] = _EventWork(done, self.send[slot].device)

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

    def close(self) -> None:
        """Releases the GPU-started exchange, if any (before the communicator is closed)."""
        if self.device_started is not None:
            self.device_started.close()
            self.device_started = None
