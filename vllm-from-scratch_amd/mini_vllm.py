"""Drop-in `mini_vllm` module: the reference engine's public names and signatures
(05_mini_vllm/mini_vllm.py:32-783) with the attention, KV-write and sampling steps
running as HIP kernels on MI355X.

    BlockAllocator, Request, ScheduledRequest, Scheduler, Sampler,
    MiniTransformer, MiniVLLM   (+ the nn building blocks they are made of)

What stays host Python: block allocation, scheduling and slot arithmetic (integer
logic, reproduced decision for decision -- see the golden scheduler traces in
tests/).  What stays torch: the dense projections (nn.Linear), RMSNorm, RoPE.
What runs in libmi355_paged.so: the KV slot write, causal prompt attention, paged
decode attention through block tables, greedy argmax and temperature/top-p
sampling.  Host-resident tensors are staged through HIP device 0 per call; there
is no CPU implementation of those steps.

Attention semantics.  The reference's decode branch reads flat KV slot 0 only and
its chunked prefill attends within the current chunk only
(05_mini_vllm/mini_vllm.py:460-481, :437-458).  This engine by default attends
every token to its true context through the request's block table; constructing
`MiniVLLM(..., reference_compat=True)` or calling the model without block tables
reproduces the reference's behaviour for result comparison.
"""
from __future__ import annotations

import math
import time
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mi355_paged import ops

_NUMPY_DTYPE = {torch.int64: np.int64, torch.int32: np.int32, torch.float32: np.float32, torch.bool: np.bool_}


def _upload(values, dtype: torch.dtype, device) -> torch.Tensor:
    """Small host data (lists, arrays) -> a device tensor without stalling the device.

    torch.tensor(values, device=...) copies from pageable memory, which the runtime does
    synchronously: the call returns only after everything queued on the stream before it has
    run, so a step that builds a dozen small index tensors alternates host and device work
    instead of overlapping them.  Staged through pinned memory (recycled by torch's host
    allocator once the copy has executed) the copy is asynchronous."""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(values, dtype=_NUMPY_DTYPE[dtype])))
    device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


# ---------------------------------------------------------------------------
# KV block allocator (host integers)
# ---------------------------------------------------------------------------

class BlockAllocator:
    """Fixed pool of physical KV blocks with reference counts, an LRU-ordered free
    list and a hash -> block map for prefix reuse (05_mini_vllm/mini_vllm.py:32-108).

    allocate() takes the least recently freed block; free() appends at the recent
    end; a block freed with a hash keeps it until the block is handed out again.
    Out of blocks -> allocate returns None (no exception).
    """

    def __init__(self, num_blocks: int, block_size: int):
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.ref_count = [0] * num_blocks
        # block id -> logical time it became free; insertion order is the eviction order
        self.free_blocks: "OrderedDict[int, float]" = OrderedDict((b, 0.0) for b in range(num_blocks))
        self.hash_to_block: dict = {}
        self.block_to_hash: dict = {}
        self._time = 0

    def _tick(self) -> float:
        self._time += 1
        return float(self._time)

    def allocate(self, block_hash: Optional[int] = None) -> Optional[int]:
        cached = self.hash_to_block.get(block_hash) if block_hash is not None else None
        if cached is not None:
            self.ref_count[cached] += 1
            self.free_blocks.pop(cached, None)   # a cached block may be sitting in the free list
            return cached
        if len(self.free_blocks) == 0:
            return None
        victim, _ = self.free_blocks.popitem(last=False)
        stale = self.block_to_hash.pop(victim, None)
        if stale is not None:
            self.hash_to_block.pop(stale, None)
        self.ref_count[victim] = 1
        return victim

    def free(self, block_id: int):
        if self.ref_count[block_id] <= 0:
            return
        self.ref_count[block_id] -= 1
        if self.ref_count[block_id] == 0:
            self.free_blocks[block_id] = self._tick()

    def mark_cached(self, block_id: int, block_hash: int):
        if block_hash not in self.hash_to_block:
            self.hash_to_block[block_hash] = block_id
            self.block_to_hash[block_id] = block_hash

    @property
    def num_free(self) -> int:
        return len(self.free_blocks)


# ---------------------------------------------------------------------------
# requests and scheduler (host integers)
# ---------------------------------------------------------------------------

@dataclass
class Request:
    """One generation request and its running state (05_mini_vllm/mini_vllm.py:115-152)."""
    request_id: str
    prompt_token_ids: List[int]
    max_new_tokens: int
    temperature: float = 1.0
    top_p: float = 1.0

    output_token_ids: List[int] = field(default_factory=list)
    num_computed_tokens: int = 0
    block_table: List[int] = field(default_factory=list)
    is_finished: bool = False
    arrival_time: float = field(default_factory=time.time)
    # prefix caching (only used when the scheduler enables it)
    block_hashes: List[int] = field(default_factory=list)   # chained hash of every full prompt block
    num_cached_tokens: int = 0                                # prompt tokens served from cached blocks
    num_registered_blocks: int = 0                            # prompt blocks already offered to the cache

    @property
    def prompt_len(self) -> int:
        return len(self.prompt_token_ids)

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_token_ids) + len(self.output_token_ids)

    @property
    def is_prefill(self) -> bool:
        return self.num_computed_tokens < len(self.prompt_token_ids)

    @property
    def next_tokens_to_compute(self) -> List[int]:
        if self.is_prefill:
            return self.prompt_token_ids[self.num_computed_tokens:]
        return self.output_token_ids[-1:]


class ScheduledRequest:
    """What one request contributes to a step: its tokens, their positions and KV slots
    (05_mini_vllm/mini_vllm.py:155-161: request, tokens, positions, is_prefill, slot_mapping).

    The tokens always sit at consecutive positions start .. start + len(tokens) - 1, so the
    `positions` and `slot_mapping` lists are only materialised when somebody reads them (the
    engine works from `start` and the request's block table with array arithmetic instead);
    a block table only ever grows at its end, so a late read gives the values of schedule time.
    """
    __slots__ = ('request', 'tokens', 'is_prefill', 'start', 'block_size', '_positions', '_slot_mapping')

    def __init__(self, request: Request, tokens: List[int], positions: Optional[List[int]] = None,
                 is_prefill: bool = False, slot_mapping: Optional[List[int]] = None,
                 start: Optional[int] = None, block_size: Optional[int] = None):
        self.request, self.tokens, self.is_prefill = request, tokens, is_prefill
        self._positions, self._slot_mapping = positions, slot_mapping
        self.start = start if start is not None else (positions[0] if positions else 0)
        self.block_size = block_size

    @property
    def positions(self) -> List[int]:
        if self._positions is None:
            self._positions = list(range(self.start, self.start + len(self.tokens)))
        return self._positions

    @property
    def slot_mapping(self) -> List[int]:
        if self._slot_mapping is None:
            bs, table = self.block_size, self.request.block_table
            self._slot_mapping = [table[pos // bs] * bs + pos % bs
                                  for pos in range(self.start, self.start + len(self.tokens))]
        return self._slot_mapping

    def __repr__(self) -> str:
        return (f"ScheduledRequest(request={self.request.request_id!r}, tokens={self.tokens!r}, "
                f"positions={self.positions!r}, is_prefill={self.is_prefill!r}, slot_mapping={self.slot_mapping!r})")


def prompt_block_hashes(token_ids: Sequence[int], block_size: int) -> List[int]:
    """One hash per FULL block of the prompt, each chained to the hash of the block before it,
    so equal hashes mean equal prefixes up to and including that block
    (06_global_prefix_cache/global_kv_pool.py:471-500: hash((parent, tokens, extra)), seed 0)."""
    hashes: List[int] = []
    parent = 0
    for start in range(0, len(token_ids) - block_size + 1, block_size):
        parent = hash((parent, tuple(token_ids[start:start + block_size]), None))
        hashes.append(parent)
    return hashes


class Scheduler:
    """First-come-first-served continuous batching with optional chunked prefill
    (05_mini_vllm/mini_vllm.py:164-337).

    Each schedule() call first revisits the running requests in order (a decode
    request costs one token; a request still in prefill takes the next chunk), then
    admits waiting requests in arrival order, all under a per-step token budget and
    a cap on the number of sequences.  Two behaviours of the reference are kept on
    purpose: a request admitted from the waiting queue with chunking disabled is
    scheduled whole even when it exceeds the remaining budget, and blocks obtained
    by a partially successful reservation stay with the request.

    enable_prefix_caching (off by default; the reference's allocator carries the hash -> block
    map, 05_mini_vllm/mini_vllm.py:62-104, but its scheduler never consults it): a request leaving
    the waiting queue first adopts, through allocate(block_hash), the cached blocks matching the
    longest prefix of its full prompt blocks and starts computing after them; every prompt block
    is offered to the cache (mark_cached) once its K/V rows have been written.  At least one
    prompt token is always computed, so a fully cached prompt re-computes its last block.
    """

    def __init__(self, max_num_seqs: int, max_num_batched_tokens: int, block_size: int,
                 allocator: BlockAllocator, enable_chunked_prefill: bool = True,
                 enable_prefix_caching: bool = False):
        self.enable_prefix_caching = enable_prefix_caching
        self.prefix_stats = {'requests': 0, 'prompt_tokens': 0, 'cached_tokens': 0}
        self.max_num_seqs = max_num_seqs
        self.max_num_batched_tokens = max_num_batched_tokens
        self.block_size = block_size
        self.allocator = allocator
        self.enable_chunked_prefill = enable_chunked_prefill
        self.waiting: List[Request] = []
        self.running: List[Request] = []
        self.finished: List[Request] = []
        # steps issued to the device whose tokens have not been handed out yet (see advance()):
        # counted by has_unfinished, so a driver loop keeps stepping until they are returned
        self.in_flight = 0
        self.dropped_items = 0      # items issued for a request that had already ended (their rows are wasted)

    def add_request(self, request: Request):
        self.waiting.append(request)

    def _ensure_blocks(self, req: Request, num_new_tokens: int) -> bool:
        """Grow req.block_table to cover num_computed_tokens + num_new_tokens."""
        needed = math.ceil((req.num_computed_tokens + num_new_tokens) / self.block_size)
        while len(req.block_table) < needed:
            got = self.allocator.allocate()
            if got is None:
                return False
            req.block_table.append(got)
        return True

    def _adopt_cached_prefix(self, req: Request):
        """Point the head of req.block_table at cached blocks holding the same prompt prefix."""
        bs = self.block_size
        req.block_hashes = prompt_block_hashes(req.prompt_token_ids, bs)
        usable = min(len(req.block_hashes), (req.prompt_len - 1) // bs)   # leave >= 1 token to compute
        for block_hash in req.block_hashes[:usable]:
            if block_hash not in self.allocator.hash_to_block:
                break
            req.block_table.append(self.allocator.allocate(block_hash))
        hit = len(req.block_table)
        req.num_computed_tokens = req.num_cached_tokens = hit * bs
        req.num_registered_blocks = hit
        self.prefix_stats['requests'] += 1
        self.prefix_stats['prompt_tokens'] += req.prompt_len
        self.prefix_stats['cached_tokens'] += hit * bs

    def _register_prompt_blocks(self, req: Request):
        """Offer every prompt block whose rows are now all in the cache to the prefix map."""
        full = min(req.num_computed_tokens, req.prompt_len) // self.block_size
        for i in range(req.num_registered_blocks, min(full, len(req.block_hashes))):
            self.allocator.mark_cached(req.block_table[i], req.block_hashes[i])
        req.num_registered_blocks = max(req.num_registered_blocks, full)

    def _compute_slot_mapping(self, req: Request, tokens: Sequence[int], start_pos: int) -> List[int]:
        bs = self.block_size
        table = req.block_table
        return [table[pos // bs] * bs + pos % bs for pos in range(start_pos, start_pos + len(tokens))]

    def _emit(self, req: Request, tokens: List[int], start: int, prefill: bool) -> ScheduledRequest:
        return ScheduledRequest(request=req, tokens=tokens, is_prefill=prefill, start=start,
                                block_size=self.block_size)

    def schedule(self) -> List[ScheduledRequest]:
        # The loop below runs once per request per step on the host while the device idles, so it is
        # written flat: limits in locals, properties and helpers expanded where they are on the
        # common path (a decode request whose table already covers its next position).
        batch: List[ScheduledRequest] = []
        budget = self.max_num_batched_tokens
        chunked = self.enable_chunked_prefill
        max_seqs, bs = self.max_num_seqs, self.block_size

        survivors: List[Request] = []
        for req in self.running:
            if req.is_finished:
                continue
            survivors.append(req)
            done = req.num_computed_tokens
            left = len(req.prompt_token_ids) - done
            if left > 0:                                   # still in its prompt: the next chunk
                want = min(left, budget) if chunked else left
                if want <= 0 or len(batch) >= max_seqs or budget < 1:
                    continue
                want = min(want, budget)          # a running request never exceeds the budget
                if not self._ensure_blocks(req, want):
                    continue
                batch.append(ScheduledRequest(req, req.prompt_token_ids[done:done + want], None, True, None, done, bs))
                budget -= want
            else:
                if len(batch) >= max_seqs or budget < 1:
                    continue
                # (the table usually covers the next position already: skip the call then)
                if len(req.block_table) * bs <= done and not self._ensure_blocks(req, 1):
                    continue
                last = req.output_token_ids[-1:] or req.prompt_token_ids[-1:]
                batch.append(ScheduledRequest(req, last, None, False, None, done, bs))
                budget -= len(last)
        self.running = survivors

        still_waiting: List[Request] = []
        for k, req in enumerate(self.waiting):
            if len(batch) >= max_seqs or budget < 1:   # room never comes back within a step:
                still_waiting.extend(self.waiting[k:])  # everybody behind stays queued, in order
                break
            if self.enable_prefix_caching and not req.block_table and req.num_computed_tokens == 0:
                self._adopt_cached_prefix(req)
            first = req.num_computed_tokens           # 0 unless a cached prefix was adopted
            left = req.prompt_len - first
            want = min(left, budget) if chunked else left
            if want > 0 and self._ensure_blocks(req, want):
                batch.append(self._emit(req, req.prompt_token_ids[first:first + want], first, True))
                budget -= want
                self.running.append(req)
            else:
                still_waiting.append(req)
        self.waiting = still_waiting
        return batch

    def update(self, scheduled: List[ScheduledRequest], next_token_ids: List[int]):
        """Advance every scheduled request by the tokens it was fed; a request whose prompt
        is fully consumed takes the sampled token, and finishes on max_new_tokens or on
        token id 0 (the demo's end-of-sequence id), returning its blocks."""
        caching = self.enable_prefix_caching
        any_finished = False
        for item, token in zip(scheduled, next_token_ids):
            req = item.request
            req.num_computed_tokens += len(item.tokens)
            if item.is_prefill:
                if caching:
                    self._register_prompt_blocks(req)
                if req.num_computed_tokens < len(req.prompt_token_ids):
                    continue                              # more prompt chunks to come
            out = req.output_token_ids
            out.append(token)
            if len(out) >= req.max_new_tokens or token == 0:
                req.is_finished = True
                free = self.allocator.free
                for block in req.block_table:
                    free(block)
                any_finished = True
                self.finished.append(req)
        if any_finished:
            self.running = [r for r in self.running if not r.is_finished]

    # update() in two halves (no counterpart in the reference), for an engine that issues step
    # i + 1 before it has read the tokens sampled by step i.  Everything schedule() looks at --
    # positions, prompt progress, output counts, finishing by length, free blocks -- is known
    # without the token values; only finishing on the end-of-sequence id needs them.

    def advance(self, scheduled: List[ScheduledRequest]) -> List[bool]:
        """The half of update() that does not need the sampled ids.  Request j of the step, if it
        takes a token, gets the stand-in -1 - j appended to its outputs (row j of the step's id
        vector: a later step feeds it from device memory) and finishes here if that reaches
        max_new_tokens.  Items whose request has finished meanwhile (it sampled the end-of-sequence
        id in the step before, found out after this one was issued) are skipped.
        Returns, per item, whether it takes a token."""
        caching = self.enable_prefix_caching
        takes = [False] * len(scheduled)
        any_finished = False
        for j, item in enumerate(scheduled):
            req = item.request
            if req.is_finished:
                self.dropped_items += 1
                continue
            req.num_computed_tokens += len(item.tokens)
            if item.is_prefill:
                if caching:
                    self._register_prompt_blocks(req)
                if req.num_computed_tokens < len(req.prompt_token_ids):
                    continue
            takes[j] = True
            out = req.output_token_ids
            out.append(-1 - j)
            if len(out) >= req.max_new_tokens:
                req.is_finished = True
                free = self.allocator.free
                for block in req.block_table:
                    free(block)
                any_finished = True
                self.finished.append(req)
        if any_finished:
            self.running = [r for r in self.running if not r.is_finished]
        return takes

    def resolve(self, scheduled: List[ScheduledRequest], takes: List[bool], token_ids: List[int]):
        """The other half: the stand-ins left by advance() become the sampled ids, and a request
        that drew the end-of-sequence id finishes now."""
        any_finished = False
        for item, took, token in zip(scheduled, takes, token_ids):
            if not took:
                continue
            req = item.request
            req.output_token_ids[-1] = token
            if token == 0 and not req.is_finished:
                req.is_finished = True
                free = self.allocator.free
                for block in req.block_table:
                    free(block)
                any_finished = True
                self.finished.append(req)
        if any_finished:
            self.running = [r for r in self.running if not r.is_finished]

    def schedule_decode_only(self) -> Optional[List[Request]]:
        """The outcome of schedule() for the steady state of generation, without building
        ScheduledRequest objects: nothing waiting (or no sequence slot left for a waiting
        request), every running request past its prompt, and enough free blocks that no request
        would be skipped.  Returns the requests in batch
        order (each feeds its last token at position num_computed_tokens) after growing their
        block tables exactly as schedule() would, or None -- having changed nothing -- when any
        precondition fails and schedule() must decide."""
        running = self.running
        n = len(running)
        if n == 0 or n > self.max_num_seqs or n > self.max_num_batched_tokens:
            return None
        # waiting requests are only admitted while there is room for another sequence; with every
        # slot taken by a running request schedule() would leave them waiting
        if self.waiting and n < self.max_num_seqs:
            return None
        bs = self.block_size
        short = 0
        for req in running:
            done = req.num_computed_tokens
            if req.is_finished or done < len(req.prompt_token_ids) or not req.output_token_ids:
                return None
            lack = -(-(done + 1) // bs) - len(req.block_table)
            if lack > 0:
                short += lack
        if short > self.allocator.num_free:
            return None
        if short:
            for req in running:
                self._ensure_blocks(req, 1)
        return list(running)

    def update_decode_only(self, batch: List[Request], next_token_ids: List[int]):
        """update() for a batch produced by schedule_decode_only()."""
        finished = []
        for req, token in zip(batch, next_token_ids):
            req.num_computed_tokens += 1
            out = req.output_token_ids
            out.append(token)
            if len(out) >= req.max_new_tokens or token == 0:
                req.is_finished = True
                for block in req.block_table:
                    self.allocator.free(block)
                finished.append(req)
        if finished:
            self.running = [r for r in self.running if not r.is_finished]
            self.finished.extend(finished)

    @property
    def has_unfinished(self) -> bool:
        return bool(self.waiting or self.running or self.in_flight)


# ---------------------------------------------------------------------------
# per-step attention plan (host) and the kernel-backed attention step
# ---------------------------------------------------------------------------

class DeviceBlockTables:
    """Device-resident copy of the running requests' block tables: int32 [rows, width].

    A request keeps one row while it runs.  Block tables only ever grow at the end, so each
    step uploads just the entries appended since the last one (one small scatter) instead of
    re-sending every table; a pure-decode step of B sequences moves about B / block_size ints.
    Rows and width grow by doubling when the pool is exhausted.
    """

    def __init__(self, device, rows: int = 8, width: int = 8):
        self.device = device
        self.tables = torch.zeros((rows, width), dtype=torch.int32, device=device)
        self._row_of: dict = {}
        self._uploaded: dict = {}
        self._free_rows = list(range(rows - 1, -1, -1))

    def _grow(self, rows: int, width: int):
        old = self.tables
        new = torch.zeros((rows, width), dtype=torch.int32, device=self.device)
        new[:old.shape[0], :old.shape[1]] = old
        self._free_rows = list(range(rows - 1, old.shape[0] - 1, -1)) + self._free_rows
        self.tables = new

    def sync(self, requests: Sequence["Request"]) -> List[int]:
        """Make sure every request has a row holding its current block table; returns the rows."""
        need_width = max((len(r.block_table) for r in requests), default=0)
        missing = sum(1 for r in requests if r.request_id not in self._row_of)
        rows, width = self.tables.shape
        if missing > len(self._free_rows) or need_width > width:
            while missing > len(self._free_rows) + (rows - self.tables.shape[0]):
                rows *= 2
            while need_width > width:
                width *= 2
            self._grow(rows, width)
        out, r_idx, c_idx, vals = [], [], [], []
        for req in requests:
            row = self._row_of.get(req.request_id)
            if row is None:
                row = self._free_rows.pop()
                self._row_of[req.request_id] = row
                self._uploaded[req.request_id] = 0
            done = self._uploaded[req.request_id]
            table = req.block_table
            if len(table) > done:
                r_idx.extend([row] * (len(table) - done))
                c_idx.extend(range(done, len(table)))
                vals.extend(table[done:])
                self._uploaded[req.request_id] = len(table)
            out.append(row)
        if vals:
            packed = _upload([r_idx, c_idx, vals], torch.int32, self.device)    # (int32 indexes are accepted as they are)
            self.tables[packed[0], packed[1]] = packed[2]
        return out

    def release(self, request_id: str):
        row = self._row_of.pop(request_id, None)
        if row is not None:
            self._uploaded.pop(request_id, None)
            self._free_rows.append(row)

    def gather(self, rows: Sequence[int]) -> torch.Tensor:
        """[len(rows), width] int32 block tables for the given rows (device gather)."""
        return self.tables.index_select(0, _upload(rows, torch.int64, self.device))


@dataclass
class AttnPlan:
    """Which kernel serves which token of a flattened step.

    Prompt segments that start at position 0 go through the causal prefill kernel
    (they only need their own fresh K/V).  Decode tokens are rows of a paged-decode
    launch with their own context length and block table.  Later chunks of a chunked
    prompt (and prompts that start after a cached prefix) are "append" segments of one
    paged-prefill launch: every token attends to the cache up to its own position; where
    that kernel does not cover the model's dtype / head_dim their tokens are paged-decode
    rows instead.
    """
    total_tokens: int
    causal_index: object                      # token rows served by causal prefill: None = all, a slice when they
                                              # are consecutive (the usual case), else an int64 [Tc] device tensor
    cu_seqlens: Optional[torch.Tensor]        # int32 [nseg + 1]
    causal_tokens: int
    max_causal_len: int
    paged_index: object                       # token rows served by paged decode, same three forms
    paged_tables: Optional[torch.Tensor]      # int32 [R, max_blocks]
    paged_context: Optional[torch.Tensor]     # int32 [R]
    paged_rows: int
    max_context: int
    append_tokens: int = 0                            # tokens served by the paged-prefill kernel
    append_tables: Optional[torch.Tensor] = None      # int32 [nseg, max_blocks]
    append_meta: Optional[torch.Tensor] = None        # int32 [3, nseg]: first token row, length, first position
    max_append_len: int = 0
    max_append_context: int = 0
    # every token of the step is the one new token of a request past its prompt (row r attends to
    # paged_context[r] = its position + 1 cached rows): RoPE, cache write and attention of a layer
    # can then be ONE launch (ops.paged_decode_append with rope tables)
    pure_decode: bool = False


def build_attn_plan(positions: Sequence[int], seq_lens: Sequence[int], is_prefill: Sequence[bool],
                    block_tables: Optional[Sequence[Sequence[int]]], device,
                    reference_compat: bool = False,
                    device_tables: Optional[DeviceBlockTables] = None,
                    table_rows: Optional[Sequence[int]] = None,
                    append_ok=False) -> AttnPlan:
    """block_tables: per-request lists of block ids (host), or None for the reference's
    semantics.  When device_tables / table_rows are given (one row id per request) the paged
    rows' tables are gathered on the device instead of being rebuilt from the host lists.
    append_ok: the paged-prefill kernel covers this model, so prompt chunks that do not start
    at position 0 become append segments instead of one paged-decode row per token; a callable
    (n_q, context) -> bool decides per segment (contexts beyond what that kernel stages)."""
    causal_tok: List[int] = []
    cu = [0]
    max_len = 0
    paged_tok: List[int] = []
    paged_ctx: List[int] = []
    paged_tab: List[Sequence[int]] = []
    paged_row: List[int] = []
    app_meta: List[List[int]] = [[], [], []]
    app_tab: List[Sequence[int]] = []
    app_row: List[int] = []
    use_rows = device_tables is not None and table_rows is not None and not reference_compat
    compat = reference_compat or (block_tables is None and not use_rows)
    tok = 0
    pure_decode = not compat
    for r, (n, prefill) in enumerate(zip(seq_lens, is_prefill)):
        n_q = n if prefill else 1
        pure_decode = pure_decode and not prefill
        start = int(positions[tok]) if n > 0 else 0
        if prefill and (compat or start == 0):
            causal_tok.extend(range(tok, tok + n_q))
            cu.append(cu[-1] + n_q)
            max_len = max(max_len, n_q)
        elif compat:
            # the reference's decode branch: context = flat KV slot 0 only
            paged_tok.append(tok)
            paged_ctx.append(1)
            paged_tab.append((0,))
        elif prefill and n_q > 1 and (append_ok(n_q, start + n_q) if callable(append_ok) else append_ok):
            app_meta[0].append(tok)
            app_meta[1].append(n_q)
            app_meta[2].append(start)
            if use_rows:
                app_row.append(table_rows[r])
            else:
                app_tab.append(block_tables[r])
        elif use_rows:
            for j in range(n_q):
                paged_tok.append(tok + j)
                paged_ctx.append(int(positions[tok + j]) + 1)
                paged_row.append(table_rows[r])
        else:
            table = block_tables[r]
            for j in range(n_q):
                paged_tok.append(tok + j)
                paged_ctx.append(int(positions[tok + j]) + 1)
                paged_tab.append(table)
        tok += n_q
    total = tok

    def idx(lst):
        if len(lst) == total:
            return None
        if lst[-1] - lst[0] + 1 == len(lst):          # ascending and consecutive: a view, no gather / scatter
            return slice(lst[0], lst[-1] + 1)
        return _upload(lst, torch.int64, device)

    plan = AttnPlan(total_tokens=total, causal_index=None, cu_seqlens=None, causal_tokens=len(causal_tok),
                    max_causal_len=max_len, paged_index=None, paged_tables=None, paged_context=None,
                    paged_rows=len(paged_tok), max_context=max(paged_ctx) if paged_ctx else 0,
                    pure_decode=pure_decode and len(paged_tok) == total and total > 0)
    if causal_tok:
        plan.causal_index = idx(causal_tok)
        plan.cu_seqlens = _upload(cu, torch.int32, device)
    if paged_tok:
        plan.paged_index = idx(paged_tok)
        if paged_row:
            plan.paged_tables = device_tables.gather(paged_row)
        else:
            width = max(len(t) for t in paged_tab)
            rows = [list(t) + [0] * (width - len(t)) for t in paged_tab]
            plan.paged_tables = _upload(rows, torch.int32, device)
        plan.paged_context = _upload(paged_ctx, torch.int32, device)
    if app_meta[0]:
        plan.append_tokens = sum(app_meta[1])
        plan.max_append_len = max(app_meta[1])
        plan.max_append_context = max(p0 + n for p0, n in zip(app_meta[2], app_meta[1]))
        plan.append_meta = _upload(app_meta, torch.int32, device)
        if app_row:
            plan.append_tables = device_tables.gather(app_row)
        else:
            width = max(len(t) for t in app_tab)
            plan.append_tables = _upload([list(t) + [0] * (width - len(t)) for t in app_tab], torch.int32, device)
    return plan


def _compute_device(t: torch.Tensor) -> torch.device:
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("mini_vllm: a HIP device is required for attention / sampling "
                           "(no CPU implementation)")
    return torch.device('cuda', 0)


def paged_attention_step(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor,
                         v_cache: torch.Tensor, slot_mapping: torch.Tensor, plan: AttnPlan,
                         scale: float, kv_written: bool = False) -> torch.Tensor:
    """KV write + attention for one layer of one step.  q, k, v: [T, H, D]; caches
    [blocks, block_size, H, D].  Returns [T, H, D] on q's device.  kv_written: the caller has
    already put k and v into the caches (fused RoPE + write), only attention is left."""
    home = q.device
    dev = _compute_device(k_cache)
    staged = not k_cache.is_cuda
    kc = k_cache.to(dev) if staged else k_cache
    vc = v_cache.to(dev) if staged else v_cache
    q, k, v = q.to(dev), k.to(device=dev, dtype=kc.dtype), v.to(device=dev, dtype=kc.dtype)
    if not kv_written:
        ops.kv_write_slots(kc, vc, k, v, slot_mapping.to(dev), layout='NSHD')
    if staged:
        k_cache.copy_(kc)
        v_cache.copy_(vc)
    out = None

    def result_rows(sel):
        """Where a kernel serving the token rows `sel` writes: the whole result, a view of it, or a
        tensor of its own that is scattered afterwards."""
        nonlocal out
        if sel is None:
            out = torch.empty(q.shape, dtype=q.dtype, device=dev)
            return out
        if out is None:
            out = torch.empty(q.shape, dtype=q.dtype, device=dev)
        return out[sel] if isinstance(sel, slice) else None

    if plan.causal_tokens:
        sel = plan.causal_index
        dst = result_rows(sel)
        part = ops.prefill_causal(q if sel is None else q[sel], k if sel is None else k[sel],
                                  v if sel is None else v[sel], plan.cu_seqlens, plan.max_causal_len, scale, out=dst)
        if dst is None:
            out[sel] = part
    if plan.paged_rows:
        sel = plan.paged_index
        dst = result_rows(sel)
        part = ops.paged_decode(q if sel is None else q[sel], kc, vc, plan.paged_tables,
                                plan.paged_context, plan.max_context, layout='NSHD', scale=scale, out=dst)
        if dst is None:
            out[sel] = part
    if plan.append_tokens:
        if out is None:
            out = torch.empty(q.shape, dtype=q.dtype, device=dev)
        meta = plan.append_meta
        ops.paged_prefill(q, kc, vc, plan.append_tables, meta[0], meta[1], meta[2], plan.max_append_len,
                          plan.max_append_context, layout='NSHD', scale=scale, out=out)
    return out.to(home)


# ---------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------

def _kernel_rows(x: torch.Tensor, *params: torch.Tensor) -> bool:
    """True when a [T, width] activation can go through the fused elementwise kernels: it
    lives on the HIP device in a supported dtype shared by the given parameters, rows are
    16-byte aligned, and nothing needs autograd (the kernels are inference-only)."""
    if not x.is_cuda or x.dim() != 2 or x.dtype not in ops.RMSNORM_MAX_HIDDEN:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        return False
    per16 = 16 // x.element_size()
    return (x.stride(1) == 1 and x.shape[1] % per16 == 0 and x.stride(0) % per16 == 0
            and all(p.dtype == x.dtype and p.device == x.device for p in params))


class RMSNorm(nn.Module):
    """weight * x / sqrt(mean(x^2) + eps) (05_mini_vllm/mini_vllm.py:344-352); one kernel
    launch for device-resident rows, the torch expression otherwise."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def fusable(self, x: torch.Tensor) -> bool:
        return _kernel_rows(x, self.weight) and x.shape[1] <= ops.RMSNORM_MAX_HIDDEN[x.dtype]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.fusable(x):
            return ops.rmsnorm(x, self.weight, self.eps)
        rms = (x * x).mean(dim=-1, keepdim=True).add(self.eps).sqrt()
        return self.weight * (x / rms)

    def add_forward(self, x: torch.Tensor, delta: torch.Tensor):
        """(x + delta, norm(x + delta)) -- the residual add fused into the normalisation."""
        if self.fusable(x) and _kernel_rows(delta) and delta.dtype == x.dtype:
            return ops.rmsnorm(x, self.weight, self.eps, delta=delta)
        total = x + delta
        return total, self.forward(total)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    lo, hi = x.chunk(2, dim=-1)
    return torch.cat((-hi, lo), dim=-1)


class RotaryEmbedding(nn.Module):
    """cos/sin tables [max_len, dim] for rotate-half RoPE, base 10000
    (05_mini_vllm/mini_vllm.py:360-374)."""

    def __init__(self, dim: int, max_len: int = 8192, base: float = 10000.0):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        angles = torch.outer(torch.arange(max_len).float(), inv_freq)
        table = torch.cat([angles, angles], dim=-1)
        self.register_buffer('inv_freq', inv_freq)
        self.register_buffer('cos_cached', table.cos())
        self.register_buffer('sin_cached', table.sin())

    def forward(self, positions: torch.Tensor):
        return self.cos_cached[positions], self.sin_cached[positions]


class SelfAttention(nn.Module):
    """Multi-head attention over a paged KV cache (05_mini_vllm/mini_vllm.py:377-487).
    The projections and RoPE are torch ops; the KV write and both attention branches
    are kernel launches driven by an AttnPlan."""

    def __init__(self, hidden_size: int, num_heads: int, head_dim: int):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(hidden_size, 3 * num_heads * head_dim, bias=False)
        self.out_proj = nn.Linear(num_heads * head_dim, hidden_size, bias=False)
        self.rope = RotaryEmbedding(head_dim)
        self.fuse_decode_step = True     # pure-decode steps: RoPE + cache write + attention in one launch

    def forward(self, x, positions, kv_cache, slot_mapping, seq_lens, is_prefill, plan: Optional[AttnPlan] = None):
        T, H, D = x.shape[0], self.num_heads, self.head_dim
        if plan is None:   # called the reference's way: no block tables -> reference semantics
            plan = build_attn_plan(positions.tolist(), seq_lens, is_prefill, None, _compute_device(kv_cache[0]))
        qkv = self.qkv(x)
        q, k, v = (t.view(T, H, D) for t in qkv.chunk(3, dim=-1))
        k_cache, v_cache = kv_cache
        cos_t, sin_t = self.rope.cos_cached, self.rope.sin_cached
        per32 = 32 // x.element_size()      # the fused kernel moves two 16-byte pieces per half row
        fusable = (_kernel_rows(qkv, cos_t, sin_t) and k_cache.is_cuda and k_cache.dtype == x.dtype and D % per32 == 0
                   and positions.dtype == torch.int64 and slot_mapping.dtype == torch.int64)
        if fusable and plan.pure_decode and self.fuse_decode_step and ops.paged_decode_append_rope_serves(x.dtype, H, H, D):
            # a step of decode tokens only: RoPE of q and k, the cache write and attention in ONE launch
            # (the new token of row r sits at position paged_context[r] - 1, where its K/V row belongs)
            attn = ops.paged_decode_append(q, k, v, k_cache, v_cache, plan.paged_tables, plan.paged_context,
                                           plan.max_context, layout='NSHD', scale=self.scale, rope=(cos_t, sin_t))
        elif fusable:
            # RoPE on q and k in place inside the projection output + cache write, one launch
            ops.rope_kv_write(q, k, v, cos_t, sin_t, positions, slot_mapping, k_cache, v_cache, layout='NSHD')
            attn = paged_attention_step(q, k, v, k_cache, v_cache, slot_mapping, plan, self.scale, kv_written=True)
        else:
            cos, sin = self.rope(positions)
            cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
            attn = paged_attention_step(q, k, v, k_cache, v_cache, slot_mapping, plan, self.scale)
        return self.out_proj(attn.reshape(plan.total_tokens, H * D).to(x.dtype))


class FFN(nn.Module):
    """SwiGLU feed-forward."""

    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.gate = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.up = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down = nn.Linear(intermediate_size, hidden_size, bias=False)
        self._gate_up = None

    def _gate_up_weight(self) -> torch.Tensor:
        """[gate.weight; up.weight] as one matrix, so that the two projections are one GEMM.
        There is no second copy of the weights to go stale: the two parameters are re-pointed at
        the halves of the fused matrix (`param.data = view`), so writes through `.data`,
        `load_state_dict` and optimiser steps land in it directly.  A parameter that was replaced
        (module `.to()`, assignment of a new Parameter) no longer points into the matrix, which is
        then rebuilt from the current values."""
        g, u = self.gate.weight, self.up.weight
        f = self._gate_up
        inter = g.shape[0]
        if (f is None or f.dtype != g.dtype or f.device != g.device or g.data_ptr() != f.data_ptr()
                or u.data_ptr() != f[inter:].data_ptr()):
            f = torch.cat([g.data, u.data], dim=0)
            g.data, u.data = f[:inter], f[inter:]
            self._gate_up = f
        return f

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _kernel_rows(x, self.gate.weight, self.up.weight):
            both = F.linear(x, self._gate_up_weight())
            inter = self.gate.weight.shape[0]
            gate, up = both[:, :inter], both[:, inter:]
            if _kernel_rows(gate) and _kernel_rows(up):
                return self.down(ops.silu_mul(gate, up))
            return self.down(F.silu(gate) * up)
        return self.down(F.silu(self.gate(x)) * self.up(x))


class TransformerLayer(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, intermediate_size: int):
        super().__init__()
        self.attn = SelfAttention(hidden_size, num_heads, hidden_size // num_heads)
        self.ffn = FFN(hidden_size, intermediate_size)
        self.norm1 = RMSNorm(hidden_size)
        self.norm2 = RMSNorm(hidden_size)

    def forward(self, x, positions, kv_cache, slot_mapping, seq_lens, is_prefill, plan=None):
        attn = self.attn(self.norm1(x), positions, kv_cache, slot_mapping, seq_lens, is_prefill, plan)
        x, normed = self.norm2.add_forward(x, attn)
        return x + self.ffn(normed)

    def forward_carry(self, x, delta, positions, kv_cache, slot_mapping, seq_lens, is_prefill, plan):
        """The same layer on a residual stream given as x + delta (delta None: just x), returning
        (stream before the feed-forward output, feed-forward output): the last addition is left to
        the next normalisation, which fuses it."""
        if delta is None:
            normed = self.norm1(x)
        else:
            x, normed = self.norm1.add_forward(x, delta)
        attn = self.attn(normed, positions, kv_cache, slot_mapping, seq_lens, is_prefill, plan)
        x, normed = self.norm2.add_forward(x, attn)
        return x, self.ffn(normed)


class MiniTransformer(nn.Module):
    """Small pre-norm transformer that owns its paged KV buffers
    k_cache / v_cache [layers, blocks, block_size, heads, head_dim]
    (05_mini_vllm/mini_vllm.py:517-585)."""

    def __init__(self, vocab_size: int, hidden_size: int, num_layers: int, num_heads: int,
                 intermediate_size: int, num_kv_blocks: int, block_size: int):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.block_size = block_size
        head_dim = hidden_size // num_heads
        self.embedding = nn.Embedding(vocab_size, hidden_size)
        self.layers = nn.ModuleList(
            TransformerLayer(hidden_size, num_heads, intermediate_size) for _ in range(num_layers))
        self.norm = RMSNorm(hidden_size)
        self.lm_head = nn.Linear(hidden_size, vocab_size, bias=False)
        cache_shape = (num_layers, num_kv_blocks, block_size, num_heads, head_dim)
        self.register_buffer('k_cache', torch.zeros(cache_shape))
        self.register_buffer('v_cache', torch.zeros(cache_shape))

    def forward(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                seq_lens: List[int], is_prefill: List[bool],
                block_tables: Optional[Sequence[Sequence[int]]] = None,
                positions_host: Optional[Sequence[int]] = None,
                reference_compat: bool = False,
                device_tables: Optional[DeviceBlockTables] = None,
                table_rows: Optional[Sequence[int]] = None) -> torch.Tensor:
        """Logits [batch, vocab] for the last token of every request in the flattened step.

        The five leading arguments are the reference's; block_tables (one list of physical
        block ids per request) enables attention over the true paged context, positions_host
        is an optional host copy of `positions` that saves a device read-back; device_tables +
        table_rows (what MiniVLLM passes) replace block_tables by rows of a persistent device copy."""
        home = self.embedding.weight.device
        token_ids, positions, slot_mapping = (t.to(home) for t in (token_ids, positions, slot_mapping))
        if positions_host is None:
            positions_host = positions.tolist()
        head_dim = self.k_cache.shape[-1]
        append_ok = self.k_cache.is_cuda and ops.paged_prefill_supported(self.k_cache.dtype, head_dim)
        if append_ok:
            dtype, bs = self.k_cache.dtype, self.block_size
            append_ok = lambda n_q, context: ops.paged_prefill_serves(dtype, head_dim, bs, n_q, context)
        plan = build_attn_plan(positions_host, seq_lens, is_prefill, block_tables,
                               _compute_device(self.k_cache), reference_compat, device_tables, table_rows,
                               append_ok=append_ok)
        last, offset = [], 0
        for n in seq_lens:
            offset += n
            last.append(offset - 1)
        return self.forward_planned(token_ids, positions, slot_mapping, plan, last, seq_lens, is_prefill)

    def forward_planned(self, token_ids: torch.Tensor, positions: torch.Tensor, slot_mapping: torch.Tensor,
                        plan: AttnPlan, last: Optional[List[int]] = None, seq_lens: Optional[List[int]] = None,
                        is_prefill: Optional[List[bool]] = None) -> torch.Tensor:
        """The layer stack for an already planned step; `last` = flattened index of each
        request's final token (None: every token is one, as in a pure decode step)."""
        x, delta = self.embedding(token_ids), None
        for i, layer in enumerate(self.layers):
            x, delta = layer.forward_carry(x, delta, positions, (self.k_cache[i], self.v_cache[i]), slot_mapping,
                                           seq_lens, is_prefill, plan)
        if isinstance(last, torch.Tensor):           # device-side row selection (captured steps)
            x = x.index_select(0, last)
            delta = delta.index_select(0, last) if delta is not None else None
        elif last is not None and len(last) != x.shape[0]:
            x = x[last]
            delta = delta[last] if delta is not None else None
        if delta is None:
            return self.lm_head(self.norm(x))
        return self.lm_head(self.norm.add_forward(x, delta)[1])

    def weights_version(self) -> tuple:
        """Addresses of every parameter and buffer a step reads.  A captured step holds these
        addresses; it stays valid under in-place edits of the values (a replay reads memory as it
        is then) and must be dropped when any tensor is replaced by another."""
        for layer in self.layers:
            # (whether or not the parameters ask for gradients: the engine's steps run without autograd
            # and would fuse at their first forward anyway -- after this key was taken, so that the
            # first captured step would be dropped as stale and captured again)
            if layer.ffn.gate.weight.is_cuda:
                layer.ffn._gate_up_weight()          # fuses, or re-fuses if a half was replaced (new addresses)
        slots = self.__dict__.get('_tensor_slots')
        if slots is None:     # (holding dict, name) of every parameter and buffer: walked once, read every step
            slots = [(d, n) for m in self.modules() for d in (m._parameters, m._buffers) for n in d
                     if d[n] is not None]
            self.__dict__['_tensor_slots'] = slots
        return tuple([d[n].data_ptr() for d, n in slots])


# ---------------------------------------------------------------------------
# captured decode step
# ---------------------------------------------------------------------------

class DecodeStepGraph:
    """A pure-decode model step recorded once as a HIP graph and replayed per step.

    The step's only per-call inputs are three int32 vectors (last token, position and
    block-table row of every sequence) staged through one pinned host buffer; slot mapping,
    context lengths and the gathered block tables are derived on the device inside the graph,
    so a replay costs one small host-to-device copy and one graph launch.  The batch is padded
    to `rows` sequences (padding rows have position -1: context 0, no cache write) and the
    attention launch is sized for contexts up to `max_context`.
    """

    def __init__(self, model: "MiniTransformer", tables: DeviceBlockTables, rows: int, max_context: int,
                 prev_ids: Optional[torch.Tensor] = None):
        dev = tables.device
        self.model, self.tables, self.rows, self.max_context = model, tables, rows, max_context
        self.table_tensor = tables.tables            # the graph is bound to this storage
        self.prev_ids = prev_ids                     # and to this one (token entries -1-j read prev_ids[j])
        # two staging buffers: a step may be issued while the previous one, which read the other
        # buffer, has not started on the device yet
        self.host = [torch.zeros((3, rows), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = torch.zeros((3, rows), dtype=torch.int32, device=dev)
        self.dev[1].fill_(-1)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up: workspaces, GEMM plans
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.logits = self._body()

    def _body(self) -> torch.Tensor:
        tok, pos, slots, ctx, tables = ops.decode_step_inputs(self.dev, self.table_tensor, self.model.block_size,
                                                              self.prev_ids)
        plan = AttnPlan(total_tokens=self.rows, causal_index=None, cu_seqlens=None, causal_tokens=0,
                        max_causal_len=0, paged_index=None, paged_tables=tables, paged_context=ctx,
                        paged_rows=self.rows, max_context=self.max_context, pure_decode=True)
        return self.model.forward_planned(tok, pos, slots, plan)

    def run(self, tokens: Sequence[int], positions: Sequence[int], rows: Sequence[int],
            parity: int = 0) -> torch.Tensor:
        """Logits [len(tokens), vocab] (a view of the graph's output buffer, valid until the next run).
        `parity` picks the staging buffer; a caller that issues a step before the previous one has
        finished alternates it."""
        n = len(tokens)
        buf = self.host_np[parity & 1]
        buf[0, :n] = tokens
        buf[1, :n] = positions
        buf[2, :n] = rows
        buf[0, n:] = 0
        buf[1, n:] = -1
        buf[2, n:] = 0
        self.dev.copy_(self.host[parity & 1], non_blocking=True)
        self.graph.replay()
        return self.logits[:n]


def _bucket_tokens(n: int) -> int:
    """Padded token count of a step: multiples of 64 up to 512, of 256 up to 4096, of 1024 above."""
    q = 64 if n <= 512 else 256 if n <= 4096 else 1024
    return max(q, -(-n // q) * q)


def _bucket_pow(n: int, base: int, floor: int) -> int:
    b = floor
    while b < n:
        b *= base
    return b


class MixedStep:
    """A step that contains prompt chunks (with or without decode tokens) at fixed, padded sizes,
    so that it can be recorded once as a HIP graph and replayed: `tokens` token rows, `slots`
    request slots, prompt chunks of at most `max_q` tokens, contexts of at most `max_ctx`.

    Token rows are laid out decode tokens first (row j = j-th decode request), prompt chunks after
    them, padding last.  Everything the kernels need comes from ONE int32 vector staged through
    pinned memory:
        [0, 3T)        token id, position (-1 = padding row: no cache write), block-table row   per token row
        [3T, 3T + 7S)  per request slot: chunk's first token row, chunk length (0 = none), chunk's first
                       position, chunk's table row | decode context length (0 = none), decode table
                       row | row of the request's last token (whose logits are wanted)
    Slot mapping and the table gathers happen on the device.  Decode rows go through the paged
    decode kernel (rows [0, S): the ones that are not decode rows have context 0, and those that are
    really prompt tokens are then overwritten), prompt chunks -- first chunks included, their K/V
    rows are in the cache by then -- through one paged-prefill launch whose grid is sized for the
    bounds and whose workgroups read their segment's actual length from device memory.
    With capture=False the same body runs eagerly (same kernels, same launch geometry: an engine
    without graphs computes bit for bit what the captured engine computes)."""

    def __init__(self, model: "MiniTransformer", tables: DeviceBlockTables, tokens: int, slots: int, max_q: int,
                 max_ctx: int, capture: bool, prev_ids: Optional[torch.Tensor] = None):
        dev = tables.device
        self.model, self.T, self.S, self.max_q, self.max_ctx = model, tokens, slots, max_q, max_ctx
        self.table_tensor = tables.tables
        self.prev_ids = prev_ids                     # token entries -1-j read prev_ids[j]
        n = 3 * tokens + 7 * slots
        # two staging buffers: a step may be filled in while the previous one, which read the other
        # buffer, has not started on the device yet
        self.host = [torch.zeros((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.dev[tokens:2 * tokens].fill_(-1)
        self.graph = None
        if capture:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():      # warm-up: workspaces, GEMM plans
                self._body()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.logits = self._body()

    def _body(self) -> torch.Tensor:
        T, S = self.T, self.S
        per_token = self.dev[:3 * T].view(3, T)
        per_slot = self.dev[3 * T:].view(7, S)
        tok, pos, slots, _, _ = ops.decode_step_inputs(per_token, self.table_tensor, self.model.block_size,
                                                       self.prev_ids, gather_tables=False)
        plan = AttnPlan(total_tokens=T, causal_index=None, cu_seqlens=None, causal_tokens=0, max_causal_len=0,
                        paged_index=slice(0, S), paged_tables=self.table_tensor.index_select(0, per_slot[5]),
                        paged_context=per_slot[4], paged_rows=S, max_context=self.max_ctx,
                        append_tokens=T, append_tables=self.table_tensor.index_select(0, per_slot[3]),
                        append_meta=per_slot[0:3], max_append_len=self.max_q, max_append_context=self.max_ctx)
        return self.model.forward_planned(tok, pos, slots, plan, last=per_slot[6])

    def run(self, nreq: int, parity: int = 0) -> torch.Tensor:
        """Logits [nreq, vocab] for the step described in host_np[parity] (valid until the next run)."""
        self.dev.copy_(self.host[parity & 1], non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
            return self.logits[:nreq]
        with torch.no_grad():
            return self._body()[:nreq]


# ---------------------------------------------------------------------------
# sampler
# ---------------------------------------------------------------------------

class Sampler:
    """temperature == 0 -> argmax; otherwise softmax(logits / T), optional nucleus
    filtering (top_p < 1) and one multinomial draw (05_mini_vllm/mini_vllm.py:592-634).
    A request may also carry an integer `top_k` attribute (> 0) -- not part of the
    reference -- which keeps only the k most probable tokens.

    The whole batch is sampled by at most two kernel launches and read back with a
    single device-to-host copy.  Random draws come from a counter-based generator
    seeded from torch's global CPU generator, so torch.manual_seed() makes runs
    repeatable; the draws are not the same numbers torch.multinomial would produce.
    """

    @staticmethod
    def sample(logits: torch.Tensor, requests: List[Request]) -> List[int]:
        if not requests:
            return []
        return [int(t) for t in Sampler.sample_device(logits, requests).tolist()]

    @staticmethod
    def sample_device(logits: torch.Tensor, requests: List[Request],
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The sampled ids as an int32 vector on the compute device, without a host read-back
        (written to the head of `out` when given)."""
        n = len(requests)
        dev = _compute_device(logits)
        rows = logits[:n].to(dev)
        temps = [float(r.temperature) for r in requests]
        greedy = [t == 0.0 for t in temps]
        picked = None
        if any(greedy):
            picked = ops.sample_argmax(rows, out=out if all(greedy) else None)
        if not all(greedy):
            top_p = [float(r.top_p) for r in requests]
            top_k = [int(getattr(r, 'top_k', 0) or 0) for r in requests]
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            drawn = ops.sample_topk_topp(
                rows, _upload([t if t != 0.0 else 1.0 for t in temps], torch.float32, dev),
                _upload(top_p, torch.float32, dev) if any(p < 1.0 for p in top_p) else None,
                _upload(top_k, torch.int32, dev) if any(k > 0 for k in top_k) else None,
                seed=seed, out=out if picked is None else None)
            if picked is None:
                picked = drawn
            else:
                picked = torch.where(_upload(greedy, torch.bool, dev), picked[:n], drawn[:n])
                if out is not None:
                    out[:n].copy_(picked)
        return picked[:n]


# ---------------------------------------------------------------------------
# engine
# ---------------------------------------------------------------------------

@dataclass
class _Flight:
    """A step with prompt chunks that has been issued: the scheduler's items, the pinned host vector
    its sampled ids are being copied to, and the event that marks the copy's end."""
    scheduled: List[ScheduledRequest]
    ids_host: torch.Tensor
    done: "torch.cuda.Event"


@dataclass
class _InflightDecode:
    """A decode step that has been issued: its requests in batch order, the pinned host vector its
    sampled ids are being copied to, and the event that marks the copy's end."""
    batch: List[Request]
    ids_host: torch.Tensor
    done: "torch.cuda.Event"


class MiniVLLM:
    """schedule -> flatten -> model forward -> sample -> update
    (05_mini_vllm/mini_vllm.py:641-783).  Extra keyword arguments, all optional:
    reference_compat (see module docstring), dtype (parameter / KV cache dtype), decode_graphs
    (replay pure-decode steps as captured HIP graphs), decode_lookahead (with decode_graphs: issue
    the next pure-decode step before the host has read this one's tokens, see _step_decode_only),
    pipeline_steps (with decode_lookahead: the same for steps that hold prompt chunks, see
    _step_pipelined) and enable_prefix_caching (reuse the KV blocks of equal prompt prefixes; needs true-context
    attention, so not with reference_compat)."""

    def __init__(self, vocab_size: int = 1000, hidden_size: int = 128, num_layers: int = 2,
                 num_heads: int = 4, intermediate_size: int = 512, num_kv_blocks: int = 256,
                 block_size: int = 16, max_num_seqs: int = 8, max_num_batched_tokens: int = 1024,
                 enable_chunked_prefill: bool = True, device: str = 'cpu',
                 reference_compat: bool = False, dtype: Optional[torch.dtype] = None,
                 decode_graphs: bool = True, enable_prefix_caching: bool = False,
                 decode_lookahead: bool = True, pipeline_steps: bool = True, fused_decode_step: bool = True):
        """The first eleven arguments are the reference's (05_mini_vllm/mini_vllm.py:649-662); the
        rest select this implementation's execution modes, every one of which gives every request
        the same tokens.  fused_decode_step: in a step of decode tokens only, RoPE, the cache write
        and attention of a layer are one launch (False: RoPE + cache write, then attention)."""
        self.device = torch.device(device)
        self.reference_compat = reference_compat
        # replay pure-decode steps as HIP graphs (device-resident engines only)
        self.decode_graphs = decode_graphs and self.device.type == 'cuda' and not reference_compat
        self._graphs: "OrderedDict[tuple, DecodeStepGraph]" = OrderedDict()
        # steps that hold prompt chunks run at padded sizes through MixedStep (captured when
        # decode_graphs is on); device-resident engines with true-context attention only
        self.mixed_steps = self.device.type == 'cuda' and not reference_compat
        self._mixed: "OrderedDict[tuple, MixedStep]" = OrderedDict()
        self.decode_lookahead = decode_lookahead and self.decode_graphs
        self._inflight: Optional[_InflightDecode] = None   # a decode step issued, its tokens not read yet
        self.pipeline_steps = pipeline_steps and self.decode_lookahead and self.mixed_steps
        self._flight: Optional[_Flight] = None             # the same for a step with prompt chunks
        self._deferred: Optional[List[ScheduledRequest]] = None   # scheduled ahead, could not be issued ahead
        self._launches = 0                                 # captured decode steps issued
        self._issued = 0                                   # steps issued through replayable bodies, of both kinds
        self._deferrals = 0                                # steps scheduled ahead that had to wait for the general path
        self._ahead = 0                                    # steps with prompt chunks issued before the previous one's ids were read
        self._ids_dev: Optional[torch.Tensor] = None       # where every captured step's sampler writes
        self._ids_host: List[torch.Tensor] = []
        self._ids_done: List[torch.cuda.Event] = []
        self.allocator = BlockAllocator(num_kv_blocks, block_size)
        self.scheduler = Scheduler(max_num_seqs=max_num_seqs,
                                   max_num_batched_tokens=max_num_batched_tokens,
                                   block_size=block_size, allocator=self.allocator,
                                   enable_chunked_prefill=enable_chunked_prefill,
                                   enable_prefix_caching=enable_prefix_caching and not reference_compat)
        self.model = MiniTransformer(vocab_size=vocab_size, hidden_size=hidden_size,
                                     num_layers=num_layers, num_heads=num_heads,
                                     intermediate_size=intermediate_size,
                                     num_kv_blocks=num_kv_blocks, block_size=block_size).to(self.device)
        if dtype is not None:
            self.model = self.model.to(dtype)
        for layer in self.model.layers:
            layer.attn.fuse_decode_step = fused_decode_step
        self.sampler = Sampler()
        self._req_counter = 0
        self._device_tables: Optional[DeviceBlockTables] = None   # created on first use (needs a HIP device)
        # logits [requests, vocab] of the most recent step, in batch order: a view of a buffer that
        # the next step may reuse (replayed steps write their logits in place) -- for inspection
        self.last_logits: Optional[torch.Tensor] = None

    def add_request(self, prompt_token_ids: List[int], max_new_tokens: int = 20,
                    temperature: float = 1.0, top_p: float = 1.0) -> str:
        req_id = f"req-{self._req_counter}"
        self._req_counter += 1
        self.scheduler.add_request(Request(request_id=req_id, prompt_token_ids=prompt_token_ids,
                                           max_new_tokens=max_new_tokens, temperature=temperature,
                                           top_p=top_p))
        return req_id

    MAX_GRAPHS = 32     # captured decode steps kept per engine (least recently used is dropped)
    MAX_MIXED = 48      # the same for steps with prompt chunks (more shapes: tokens x slots x chunk x context)

    def _mixed_step(self, tokens: int, nreq: int, max_q: int, max_ctx: int) -> MixedStep:
        """The (captured) step serving `tokens` token rows of `nreq` requests with prompt chunks up
        to `max_q` tokens and contexts up to `max_ctx`, at the padded sizes of its bucket."""
        S = _bucket_pow(nreq, 4, 8)           # 8, 32, 128, ...: few slot counts, so few distinct graphs
        T = max(_bucket_tokens(tokens), S)
        Q = min(_bucket_pow(max(max_q, 1), 4, 16), T)
        C = _bucket_pow(max_ctx, 2, 256)
        tables = self._device_tables
        key = (T, S, Q, C, tables.tables.data_ptr(), tables.tables.shape[1], self.model.weights_version(),
               self._ids_dev.data_ptr())
        step = self._mixed.get(key)
        if step is None:
            for stale in [k for k in self._mixed if k[4:] != key[4:]]:
                del self._mixed[stale]           # block tables, KV buffers or weights were replaced
            while len(self._mixed) >= self.MAX_MIXED:
                self._mixed.popitem(last=False)
            step = self._mixed[key] = MixedStep(self.model, tables, T, S, Q, C, capture=self.decode_graphs,
                                                prev_ids=self._ids_dev)
        else:
            self._mixed.move_to_end(key)
        return step

    def _run_mixed(self, scheduled: List[ScheduledRequest], rows: List[int], parity: int = 0) -> Optional[torch.Tensor]:
        """Logits [len(scheduled), vocab], in scheduled order, of a step through MixedStep; None when
        the step's contexts are beyond what the paged-prefill kernels serve for this chunk length
        (the general path then splits such contexts over paged-decode rows).  A decode token
        -1 - j stands for entry j of the ids the previous step sampled (read on the device)."""
        decode = [i for i, item in enumerate(scheduled) if not item.is_prefill and len(item.tokens) == 1]
        is_decode = set(decode)
        chunks = [i for i in range(len(scheduled)) if i not in is_decode]
        total = sum(len(item.tokens) for item in scheduled)
        max_q = max((len(scheduled[i].tokens) for i in chunks), default=1)
        max_ctx = max(item.start + len(item.tokens) for item in scheduled)
        cache = self.model.k_cache
        if not ops.paged_prefill_serves(cache.dtype, cache.shape[-1], self.model.block_size,
                                        _bucket_pow(max(max_q, 1), 4, 16), _bucket_pow(max_ctx, 2, 256)):
            return None
        step = self._mixed_step(total, len(scheduled), max_q, max_ctx)
        T, S = step.T, step.S
        buf = step.host_np[parity & 1]
        per_token = buf[:3 * T].reshape(3, T)
        per_slot = buf[3 * T:].reshape(7, S)
        per_token[0] = 0
        per_token[1] = -1
        per_token[2] = 0
        per_slot[:] = 0
        nd = len(decode)
        if nd:
            starts = np.fromiter((scheduled[i].start for i in decode), dtype=np.int32, count=nd)
            drows = np.fromiter((rows[i] for i in decode), dtype=np.int32, count=nd)
            per_token[0, :nd] = np.fromiter((scheduled[i].tokens[0] for i in decode), dtype=np.int32, count=nd)
            per_token[1, :nd] = starts
            per_token[2, :nd] = drows
            per_slot[4, :nd] = starts + 1
            per_slot[5, :nd] = drows
            per_slot[6, decode] = np.arange(nd, dtype=np.int32)
        at = nd
        for s_idx, i in enumerate(chunks):
            item = scheduled[i]
            req, n, start = item.request, len(item.tokens), item.start
            if item.is_prefill:
                prompt = getattr(req, '_prompt_array', None)
                if prompt is None:
                    prompt = req._prompt_array = np.asarray(req.prompt_token_ids, dtype=np.int64)
                per_token[0, at:at + n] = prompt[start:start + n]
            else:
                per_token[0, at:at + n] = item.tokens
            per_token[1, at:at + n] = np.arange(start, start + n, dtype=np.int32)
            per_token[2, at:at + n] = rows[i]
            per_slot[0, s_idx] = at
            per_slot[1, s_idx] = n
            per_slot[2, s_idx] = start
            per_slot[3, s_idx] = rows[i]
            per_slot[6, i] = at + n - 1
            at += n
        return step.run(len(scheduled), parity)

    def _ensure_device_state(self) -> None:
        """Block tables on the device, the vector every replayed step's sampler writes, and two
        pinned vectors (with their events) through which sampled ids reach the host."""
        if self._device_tables is None:
            dev = _compute_device(self.model.k_cache)
            self._device_tables = DeviceBlockTables(dev, rows=max(8, self.scheduler.max_num_seqs))
        if self._ids_dev is None:
            dev = self._device_tables.device
            cap = self.scheduler.max_num_seqs + 64          # every padded batch size fits
            self._ids_dev = torch.zeros((cap,), dtype=torch.int32, device=dev)
            self._ids_host = [torch.zeros((cap,), dtype=torch.int32).pin_memory() for _ in range(2)]
            self._ids_done = [torch.cuda.Event() for _ in range(2)]

    def _launch_mixed(self, scheduled: List[ScheduledRequest]) -> Optional[_Flight]:
        """Issues a step with prompt chunks through MixedStep, its sampling and the copy of the
        sampled ids to pinned host memory, without waiting for the device; None (nothing issued)
        when MixedStep does not serve the step."""
        self._ensure_device_state()
        requests = [item.request for item in scheduled]
        parity = self._issued & 1
        logits = self._run_mixed(scheduled, self._device_tables.sync(requests), parity)
        if logits is None:
            return None
        self._issued += 1
        self.last_logits = logits
        n = len(requests)
        ids = self.sampler.sample_device(logits, requests, out=self._ids_dev)
        self._ids_host[parity][:n].copy_(ids, non_blocking=True)
        self._ids_done[parity].record()
        return _Flight(scheduled, self._ids_host[parity], self._ids_done[parity])

    def _finish_flight(self, cur: _Flight, takes: List[bool]):
        """Waits for an issued step's ids, completes the scheduler's bookkeeping and returns the
        step's [(request_id, token)]."""
        cur.done.synchronize()
        tokens = cur.ids_host[:len(cur.scheduled)].tolist()
        self.scheduler.resolve(cur.scheduled, takes, tokens)
        out = []
        release = self._device_tables.release
        for item, took, tok in zip(cur.scheduled, takes, tokens):
            if took:
                req = item.request
                out.append((req.request_id, tok))
                if req.is_finished:
                    release(req.request_id)
        return out

    def _step_pipelined(self, cur: _Flight):
        """Completes the issued step `cur`, after issuing the step that follows it.

        The scheduler only needs a step's token VALUES to end a request on the end-of-sequence
        id; everything else (positions, prompt progress, finishing by length, blocks) follows from
        the step's shape.  So the step's bookkeeping is applied first with stand-ins for the ids
        (Scheduler.advance), the next step is scheduled and issued -- its decode rows read the ids
        from device memory -- and only then does the host wait for the ids and patch them in
        (Scheduler.resolve): scheduling, table upkeep and input staging overlap the device's work.
        A request that turns out to have drawn the end-of-sequence id has one wasted row in the
        step already issued (its result is dropped) and frees its slot one step later than the
        step-by-step loop would; each request's tokens are the same either way."""
        sch = self.scheduler
        takes = sch.advance(cur.scheduled)
        nxt = None
        # a pure-decode phase ahead (nobody waiting, nobody in its prompt) is left to the captured
        # decode steps, which cost the host far less
        if self.pipeline_steps and (sch.waiting or any(r.num_computed_tokens < len(r.prompt_token_ids)
                                                       for r in sch.running)):
            ahead = sch.schedule()
            if ahead:
                nxt = self._launch_mixed(ahead)
                if nxt is None:
                    self._deferred = ahead
                    self._deferrals += 1
                else:
                    self._ahead += 1
        self._flight = nxt
        sch.in_flight = 1 if (nxt is not None or self._deferred is not None) else 0
        return self._finish_flight(cur, takes)

    def _decode_graph(self, nseq: int, max_context: int) -> DecodeStepGraph:
        """The captured step serving `nseq` sequences with contexts up to `max_context`: batch
        padded to 8, then to a multiple of 16 / 32 / 64 (up to 64 / 256 / beyond), context bound
        rounded up to a multiple of 256."""
        quantum = 8 if nseq <= 8 else 16 if nseq <= 64 else 32 if nseq <= 256 else 64
        rows = -(-nseq // quantum) * quantum
        bound = -(-max_context // 256) * 256
        tables = self._device_tables
        key = (rows, bound, tables.tables.data_ptr(), tables.tables.shape[1], self.model.k_cache.data_ptr(),
               self.model.weights_version(), self._ids_dev.data_ptr())
        graph = self._graphs.get(key)
        if graph is None:
            for stale in [k for k in self._graphs if k[2:] != key[2:]]:
                del self._graphs[stale]          # block tables, KV buffers or weights were replaced
            while len(self._graphs) >= self.MAX_GRAPHS:
                self._graphs.popitem(last=False)
            graph = self._graphs[key] = DecodeStepGraph(self.model, tables, rows, bound, self._ids_dev)
        else:
            self._graphs.move_to_end(key)
        return graph

    def _launch_decode(self, batch: List[Request], tokens: List[int], positions: List[int]) -> "_InflightDecode":
        """Issues one captured decode step, its sampling and the copy of the sampled ids to pinned
        host memory; nothing here waits for the device."""
        self._ensure_device_state()
        rows = self._device_tables.sync(batch)
        parity = self._issued & 1
        self._issued += 1
        self._launches += 1
        n = len(batch)
        logits = self.last_logits = self._decode_graph(n, max(positions) + 1).run(tokens, positions, rows, parity)
        ids = self.sampler.sample_device(logits, batch, out=self._ids_dev)
        self._ids_host[parity][:n].copy_(ids, non_blocking=True)
        self._ids_done[parity].record()
        return _InflightDecode(batch, self._ids_host[parity], self._ids_done[parity])

    def _launch_ahead(self, cur: "_InflightDecode") -> Optional["_InflightDecode"]:
        """The decode step after `cur`, issued while `cur` is still running: every request of
        `cur` that cannot finish by length feeds the id `cur` samples for it, read on the device.
        A request that turns out to have sampled the end-of-sequence id is only known after the
        fact: its row of the step issued here is wasted and its result dropped.  None when the
        scheduler would not produce a pure-decode step of exactly these requests."""
        sch = self.scheduler
        live = [(j, r) for j, r in enumerate(cur.batch) if not r.is_finished]
        if len(live) != len(sch.running):
            return None
        ahead = [(j, r) for j, r in live if len(r.output_token_ids) + 1 < r.max_new_tokens]
        if not ahead or len(ahead) > sch.max_num_seqs or len(ahead) > sch.max_num_batched_tokens:
            return None
        # a waiting request joins as soon as a sequence slot frees: only look ahead past it while
        # every slot stays taken (nobody reaches its length limit with the step in flight)
        if sch.waiting and len(ahead) < sch.max_num_seqs:
            return None
        bs = sch.block_size
        short = 0
        for _, r in ahead:      # positions 0 .. num_computed_tokens + 1 must have blocks
            short += max(0, -(-(r.num_computed_tokens + 2) // bs) - len(r.block_table))
        if short > self.allocator.num_free:
            return None
        if short:
            for _, r in ahead:
                sch._ensure_blocks(r, 2)
        return self._launch_decode([r for _, r in ahead], [-1 - j for j, _ in ahead],
                                   [r.num_computed_tokens + 1 for _, r in ahead])

    def _step_decode_only(self):
        """A step in which every running request decodes one token, served by a captured
        graph; None when the scheduler state needs the general path.

        With decode_lookahead the step after this one is issued before this one's tokens are
        read back (its token inputs are this step's sampled ids, taken on the device), so the
        host's scheduling, table upkeep and bookkeeping overlap the device's work instead of
        alternating with it.  Results per request are those of the step-by-step loop; what
        differs is when KV blocks are taken (one token earlier) and that a sequence ending on
        the end-of-sequence id costs one wasted row in the step already in flight."""
        cur, self._inflight = self._inflight, None
        if cur is None:
            batch = self.scheduler.schedule_decode_only()
            if batch is None:
                return None
            cur = self._launch_decode(batch, [r.output_token_ids[-1] for r in batch],
                                      [r.num_computed_tokens for r in batch])
        nxt = self._launch_ahead(cur) if self.decode_lookahead else None
        cur.done.synchronize()
        ids = cur.ids_host[:len(cur.batch)].tolist()
        batch, next_tokens = [], []
        for req, tok in zip(cur.batch, ids):
            if not req.is_finished:             # finished: ended on an earlier step's end-of-sequence id
                batch.append(req)
                next_tokens.append(tok)
        self.scheduler.update_decode_only(batch, next_tokens)
        for req in batch:
            if req.is_finished:
                self._device_tables.release(req.request_id)
        if nxt is not None and any(not r.is_finished for r in nxt.batch):
            self._inflight = nxt
        return [(req.request_id, tok) for req, tok in zip(batch, next_tokens)]

    def _mixed_supported(self) -> bool:
        """MixedStep needs the paged-prefill kernels for this model's dtype and head_dim, device-
        resident weights and KV buffers, and inference-only parameters."""
        ok = self.__dict__.get('_mixed_ok')
        if ok is None:
            cache = self.model.k_cache
            ok = self._mixed_ok = (cache.is_cuda and self.model.embedding.weight.is_cuda
                                   and ops.paged_prefill_supported(cache.dtype, cache.shape[-1])
                                   and cache.dtype == self.model.embedding.weight.dtype)
        return ok

    def step(self):
        """One engine step; returns [(request_id, new_token)] for every request that
        produced a token (decode steps and the final chunk of a prompt)."""
        if self._flight is not None:
            cur, self._flight = self._flight, None
            return self._step_pipelined(cur)
        scheduled = None
        if self._deferred is not None:
            # scheduled while the step before was in flight: decode items carry stand-ins for
            # ids that have arrived since, and a request may have ended on one of them
            scheduled, self._deferred = self._deferred, None
            self.scheduler.in_flight = 0
            scheduled = [item for item in scheduled if not item.request.is_finished]
            for item in scheduled:
                if not item.is_prefill:
                    item.tokens = item.request.output_token_ids[-1:]
        elif self.decode_graphs:
            fast = self._step_decode_only()
            if fast is not None:
                return fast
        if scheduled is None:
            scheduled = self.scheduler.schedule()
        if not scheduled:
            return []
        if self.mixed_steps and self._mixed_supported():
            cur = self._launch_mixed(scheduled)
            if cur is not None:
                return self._step_pipelined(cur)
        seq_lens = [len(item.tokens) for item in scheduled]
        prefill_flags = [item.is_prefill for item in scheduled]
        requests = [item.request for item in scheduled]
        # the step's flattened (token, position, slot) vectors, by array arithmetic per request:
        # positions are consecutive from item.start, slot = block_table[pos // bs] * bs + pos % bs
        bs = self.scheduler.block_size
        host_inputs = np.empty((3, sum(seq_lens)), dtype=np.int64)
        tok_row, pos_row, slot_row = host_inputs
        at = 0
        for item, n in zip(scheduled, seq_lens):
            req = item.request
            start = item.start
            if n == 1:                                   # a decode token (or a one-token chunk): plain integers
                tok_row[at] = item.tokens[0]
                pos_row[at] = start
                slot_row[at] = req.block_table[start // bs] * bs + start % bs
            else:
                if item.is_prefill:
                    prompt = getattr(req, '_prompt_array', None)
                    if prompt is None:
                        prompt = req._prompt_array = np.asarray(req.prompt_token_ids, dtype=np.int64)
                    tok_row[at:at + n] = prompt[start:start + n]
                else:
                    tok_row[at:at + n] = item.tokens
                pos = np.arange(start, start + n, dtype=np.int64)
                pos_row[at:at + n] = pos
                table = np.asarray(req.block_table, dtype=np.int64)
                slot_row[at:at + n] = table[pos // bs] * bs + pos % bs
            at += n

        rows = None
        if not self.reference_compat:
            if self._device_tables is None:
                self._device_tables = DeviceBlockTables(_compute_device(self.model.k_cache),
                                                        rows=max(8, self.scheduler.max_num_seqs))
            rows = self._device_tables.sync(requests)
        # one host-to-device copy for the three per-token vectors (the reference makes three, :737-739)
        step_inputs = _upload(host_inputs, torch.int64, self.device)
        with torch.no_grad():
            logits = self.model(step_inputs[0], step_inputs[1], step_inputs[2], seq_lens, prefill_flags,
                                positions_host=host_inputs[1], reference_compat=self.reference_compat,
                                device_tables=self._device_tables, table_rows=rows)
        self.last_logits = logits
        next_tokens = self.sampler.sample(logits, requests)
        self.scheduler.update(scheduled, next_tokens)
        if self._device_tables is not None:
            for req in requests:
                if req.is_finished:
                    self._device_tables.release(req.request_id)
        return [(item.request.request_id, tok) for item, tok in zip(scheduled, next_tokens)
                if not (item.is_prefill and item.request.is_prefill)]

    def generate(self, prompts: List[List[int]], max_new_tokens: int = 20):
        """Run the given prompts to completion -> {request_id: generated token ids}."""
        ids = [self.add_request(p, max_new_tokens=max_new_tokens) for p in prompts]
        outputs = {rid: [] for rid in ids}
        while self.scheduler.has_unfinished:
            for rid, tok in self.step():
                if rid in outputs:
                    outputs[rid].append(tok)
        return outputs


# ---------------------------------------------------------------------------
# data-parallel front end (no counterpart in the reference, which is single-device)
# ---------------------------------------------------------------------------

class DataParallelMiniVLLM:
    """MiniVLLM.generate over the ranks of a torch.distributed group, one process per GPU.

    Sequences never interact on this path, so the batch is cut by rows: rank r serves the
    contiguous slice `dp.shard_bounds(len(prompts), world, r)` with an engine of its own (its own
    BlockAllocator, Scheduler and KV buffers; weights replicated, broadcast from rank 0 when the
    engine is built).  The only exchange is one all-gather per engine step of the ids that step
    sampled (backend "nccl" = RCCL over xGMI; "gloo" for CPU rehearsals), so that every rank --
    whichever one the front end talks to -- holds every request's tokens.  A rank's contribution
    to a step is a fixed-size int32 vector: up to max_num_seqs (local request index, token id)
    pairs, -1 padded, and a last entry saying whether the rank still has unfinished requests.
    The collective of step i runs while step i + 1 computes (dp.OverlappedTokenGather); ranks do
    not need each other's ids to go on decoding.  All ranks leave the loop on the same iteration:
    the first one whose gathered vectors say that nobody has work left.

    Under greedy sampling the result equals, token for token, what one engine returns for the whole
    list of prompts: a request's tokens depend only on its own prompt and the weights.
    """

    def __init__(self, group=None, single_rank_collective: bool = False, **engine_kwargs):
        """`single_rank_collective`: issue the per-step collective on a one-rank group too (a rehearsal of
        the RCCL path on a one-GPU machine); `engine_kwargs` are MiniVLLM's."""
        import torch.distributed as dist
        from mi355_paged import dp
        self._dp = dp
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.engine = MiniVLLM(**engine_kwargs)
        self._req_counter = 0
        backend = dist.get_backend(group) if on else None
        # the collective moves device memory with RCCL and host memory with gloo
        self._wire_device = self.engine.device if backend == 'nccl' else torch.device('cpu')
        if self.world > 1:
            self._replicate_weights(dist, backend)
        self.cap = self.engine.scheduler.max_num_seqs
        # Over RCCL the exchange runs through the library's own communicator: the upload of a step's
        # vector and its all-gather are one captured graph per buffer pair, replayed on a side
        # stream (one graph launch per engine step on the host, nothing on the compute stream).
        # Where the communicator cannot be set up, torch.distributed carries the collective.
        self._native = None
        if backend == 'nccl' and (self.world > 1 or single_rank_collective):
            try:
                self._native = dp.NativeComm(self.engine.device, group)
            except Exception:
                self._native = None
        self._gather = dp.OverlappedTokenGather(2 * self.cap + 1, self._wire_device, group=group,
                                                single_rank_collective=single_rank_collective, native=self._native,
                                                device_started=False)    # the vectors come from the host

    def _replicate_weights(self, dist, backend) -> None:
        """Every rank takes rank 0's parameters and tables (the KV buffers are per-rank state)."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        for name, t in list(self.engine.model.named_parameters()) + list(self.engine.model.named_buffers()):
            if name in ('k_cache', 'v_cache'):
                continue
            data = t.data
            if backend == 'gloo' and data.is_cuda:
                host = data.cpu()
                dist.broadcast(host, src=src, group=self.group)
                data.copy_(host)
            else:
                dist.broadcast(data, src=src, group=self.group)

    def generate(self, prompts: List[List[int]], max_new_tokens: int = 20, temperature: float = 1.0,
                 top_p: float = 1.0):
        """{request_id: generated token ids} for ALL prompts, identical on every rank; request ids
        count over the whole list ("req-0" is prompts[0]) as a single engine's would."""
        lo, hi = self._dp.shard_bounds(len(prompts), self.world, self.rank)
        bounds = [self._dp.shard_bounds(len(prompts), self.world, r) for r in range(self.world)]
        base = self._req_counter
        self._req_counter += len(prompts)
        outputs = {f"req-{base + i}": [] for i in range(len(prompts))}
        local_index = {}
        for i in range(lo, hi):
            rid = self.engine.add_request(prompts[i], max_new_tokens=max_new_tokens, temperature=temperature,
                                          top_p=top_p)
            local_index[rid] = i - lo
        gather, cap, step = self._gather, self.cap, 0

        def consume(s: int) -> bool:
            """File step s's tokens under their requests; True when no rank has work left."""
            vec = gather.result(s).view(self.world, 2 * cap + 1).cpu().tolist()
            busy = False
            for r, row in enumerate(vec):
                for j in range(cap):
                    idx, tok = row[2 * j], row[2 * j + 1]
                    if idx < 0:
                        break
                    outputs[f"req-{base + bounds[r][0] + idx}"].append(tok)
                busy = busy or row[2 * cap] != 0
            return not busy

        while True:
            results = self.engine.step() if self.engine.scheduler.has_unfinished else []
            if len(results) > cap:
                raise RuntimeError(f"a step produced {len(results)} tokens for {cap} sequence slots")
            host = gather.host_buffer(step)
            host.fill_(-1)
            for j, (rid, tok) in enumerate(results):
                host[2 * j] = local_index[rid]
                host[2 * j + 1] = tok
            host[2 * cap] = 1 if self.engine.scheduler.has_unfinished else 0
            gather.launch_from_host(step)
            # step - 1's collective has had a whole engine step to complete
            if step >= 1 and consume(step - 1):
                break
            step += 1
        gather.drain()        # the last launch carried no tokens: nobody had work when it was issued
        return outputs


def benchmark_throughput(dtype=None, warm_passes: int = 1):
    """Output tokens per second of generate() on the configuration the reference benchmarks
    (05_mini_vllm/mini_vllm.py:936-974: vocabulary 1000, hidden 256, 4 layers, 8 heads, intermediate
    1024, 256 KV blocks of 16, 16 sequence slots, 1024-token budget; 16 prompts of 10-50 random
    tokens, 20 new tokens each, temperature-1 sampling), measured as the reference measures it:
    wall clock around one generate() call.  `warm_passes` untimed calls come first (the first call
    on a fresh engine also captures its steps; 0 times the cold call, as the reference does).
    Needs a HIP device.  Prints the figures and returns output tokens per second.
    """
    import random
    import time
    engine = MiniVLLM(vocab_size=1000, hidden_size=256, num_layers=4, num_heads=8, intermediate_size=1024,
                      num_kv_blocks=256, block_size=16, max_num_seqs=16, max_num_batched_tokens=1024,
                      device='cuda', dtype=dtype)
    prompts = [[random.randint(1, 999) for _ in range(random.randint(10, 50))] for _ in range(16)]
    for _ in range(warm_passes):
        engine.generate(prompts, max_new_tokens=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outputs = engine.generate(prompts, max_new_tokens=20)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_out = sum(len(v) for v in outputs.values())
    print("benchmark_throughput:")
    print(f"  concurrent requests: {len(prompts)}")
    print(f"  input tokens: {sum(len(p) for p in prompts)}")
    print(f"  output tokens: {n_out}")
    print(f"  elapsed: {elapsed:.3f}s")
    print(f"  output throughput: {n_out / elapsed:.1f} tok/s")
    return n_out / elapsed
