"""Drop-in `paged_attention` module: the reference's public names and signatures
(01_paged_attention/paged_attention.py:12-136), computed by HIP kernels on MI355X.

    KVCachePool, compute_slot_mapping, paged_attention_decode,
    paged_attention_prefill, BLOCK_SIZE

Block bookkeeping and slot arithmetic are host integer logic and stay in Python.
The KV gather + attention and the KV writes run in libmi355_paged.so; tensors
that live on the host are staged through HIP device 0 for the call and results
are returned on the caller's device.  There is no CPU implementation: without
a HIP device the compute functions raise RuntimeError.

`paged_attention_decode_batch` is the batched form the kernels natively work
on (many sequences per launch); the single-sequence function is that with B=1.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch

from mi355_paged import ops

BLOCK_SIZE = 16  # tokens per physical block (01_paged_attention/paged_attention.py:12)

_IntList = Union[Sequence[int], torch.Tensor]


def _compute_device(*tensors: torch.Tensor) -> torch.device:
    for t in tensors:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("paged_attention: a HIP device is required (no CPU implementation)")
    return torch.device('cuda', 0)


def _as_int_list(x: _IntList) -> List[int]:
    return [int(v) for v in (x.tolist() if isinstance(x, torch.Tensor) else x)]


class KVCachePool:
    """Physical KV block pool: one tensor [2, layers, blocks, kv_heads, block_size, head_dim]
    plus a FIFO free list with reference counts (01_paged_attention/paged_attention.py:15-51)."""

    def __init__(self, num_blocks: int, block_size: int, num_kv_heads: int, head_dim: int,
                 num_layers: int, dtype=torch.float32, device='cpu'):
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.kv_cache = torch.zeros((2, num_layers, num_blocks, num_kv_heads, block_size, head_dim),
                                    dtype=dtype, device=device)
        self.ref_counts = [0] * num_blocks
        self.free_blocks: List[int] = list(range(num_blocks))

    def allocate(self, num_blocks: int) -> List[int]:
        available = len(self.free_blocks)
        if num_blocks > available:
            raise RuntimeError(f"OOM: need {num_blocks}, have {available}")
        taken = self.free_blocks[:num_blocks]
        del self.free_blocks[:num_blocks]
        for block in taken:
            self.ref_counts[block] = 1
        return taken

    def free(self, block_ids: List[int]) -> None:
        for block in block_ids:
            self.ref_counts[block] -= 1
            if self.ref_counts[block] == 0:
                self.free_blocks.append(block)

    def write_kv(self, layer_idx: int, block_id: int, slot_in_block: int,
                 key: torch.Tensor, value: torch.Tensor):
        """Store one token's K and V rows ([kv_heads, head_dim] each)."""
        cache = self.kv_cache
        dev = _compute_device(cache, key, value)
        slot = _upload([block_id * self.block_size + slot_in_block], torch.int64, dev)
        k_rows = key.to(device=dev, dtype=cache.dtype).reshape(1, self.num_kv_heads, self.head_dim)
        v_rows = value.to(device=dev, dtype=cache.dtype).reshape(1, self.num_kv_heads, self.head_dim)
        if cache.is_cuda:
            ops.kv_write_slots(cache[0, layer_idx], cache[1, layer_idx], k_rows, v_rows, slot, 'NHSD')
        else:
            # host-resident pool: stage the touched block through the device and copy it back
            staged = cache[:, layer_idx, block_id].to(dev).unsqueeze(1).contiguous()  # [2,1,H,BS,D]
            local = _upload([slot_in_block], torch.int64, dev)
            ops.kv_write_slots(staged[0], staged[1], k_rows, v_rows, local, 'NHSD')
            cache[:, layer_idx, block_id].copy_(staged[:, 0])


def _upload(values, dtype: torch.dtype, dev: torch.device) -> torch.Tensor:
    """Small host integers -> device tensor through pinned memory, without blocking (a copy from
    pageable memory waits for everything already queued on the stream)."""
    t = torch.tensor(values, dtype=dtype)
    return t.pin_memory().to(dev, non_blocking=True) if dev.type == 'cuda' else t.to(dev)


def compute_slot_mapping(block_table: List[int], seq_len: int, block_size: int) -> List[int]:
    """Flat KV slot of every token position 0..seq_len-1:
    block_table[pos // block_size] * block_size + pos % block_size
    (01_paged_attention/paged_attention.py:54-64)."""
    slots: List[int] = []
    for logical, physical in enumerate(block_table):
        first = logical * block_size
        if first >= seq_len:
            break
        base = physical * block_size
        slots.extend(range(base, base + min(block_size, seq_len - first)))
    if len(slots) < seq_len:
        raise IndexError("list index out of range")  # block table shorter than the sequence
    return slots


def paged_attention_decode_batch(
    query: torch.Tensor,          # [B, num_q_heads, head_dim]
    kv_cache: torch.Tensor,       # [2, num_blocks, num_kv_heads, block_size, head_dim]
    block_tables: torch.Tensor,   # [B, max_blocks] int32
    context_lens: torch.Tensor,   # [B] int32
    max_context: Optional[int] = None,
) -> torch.Tensor:
    """Decode attention for B independent sequences in one launch (device tensors)."""
    if max_context is None:
        max_context = int(context_lens.max().item()) if context_lens.numel() else 0
    return ops.paged_decode(query, kv_cache[0], kv_cache[1], block_tables, context_lens,
                            max_context, layout='NHSD')


def paged_attention_decode(
    query: torch.Tensor,       # [num_q_heads, head_dim]
    kv_cache: torch.Tensor,    # [2, num_blocks, num_kv_heads, block_size, head_dim]
    block_table: _IntList,     # logical block -> physical block
    seq_len: int,
    head_dim: int,
    num_q_heads: int,
    num_kv_heads: int,
    block_size: int = BLOCK_SIZE,
) -> torch.Tensor:
    """softmax(q K^T / sqrt(head_dim)) V over the seq_len cached tokens reached through
    block_table; query heads g*kv_h .. g*(kv_h+1)-1 read kv head kv_h
    (01_paged_attention/paged_attention.py:67-101).  Returns a new [num_q_heads, head_dim]
    tensor of query's dtype on query's device."""
    if seq_len <= 0:
        raise RuntimeError("stack expects a non-empty TensorList")  # what the reference raises
    if (seq_len + block_size - 1) // block_size > len(block_table):
        raise IndexError("list index out of range")
    if tuple(query.shape) != (num_q_heads, head_dim):
        raise ValueError(f"query must be [{num_q_heads}, {head_dim}], got {tuple(query.shape)}")
    if kv_cache.dim() != 5 or kv_cache.shape[0] != 2 or kv_cache.shape[2] != num_kv_heads \
            or kv_cache.shape[3] != block_size or kv_cache.shape[4] != head_dim:
        raise ValueError("kv_cache must be [2, num_blocks, num_kv_heads, block_size, head_dim]")
    dev = _compute_device(query, kv_cache)
    cache = kv_cache.to(device=dev, dtype=query.dtype)
    q = query.to(dev).unsqueeze(0)
    if isinstance(block_table, torch.Tensor) and block_table.is_cuda:
        # a table that already lives on the device is used where it is (no read-back)
        tables = block_table.to(device=dev, dtype=torch.int32).reshape(1, -1)
        lens = _upload([seq_len], torch.int32, dev)
    else:
        # block table and context length travel in one host-to-device copy
        # (through pinned memory and without blocking: a copy from pageable memory would wait for
        # everything queued on the stream, i.e. for the previous decode step)
        packed = _upload(_as_int_list(block_table) + [seq_len], torch.int32, dev)
        tables, lens = packed[:-1].unsqueeze(0), packed[-1:]
    out = ops.paged_decode(q, cache[0], cache[1], tables, lens, seq_len, layout='NHSD')
    return out[0].to(query.device)


def paged_attention_prefill(
    query: torch.Tensor,   # [seq_len, num_q_heads, head_dim]
    key: torch.Tensor,     # [seq_len, num_kv_heads, head_dim]
    value: torch.Tensor,   # [seq_len, num_kv_heads, head_dim]
    kv_cache: torch.Tensor,
    block_table: _IntList,
    block_size: int,
) -> torch.Tensor:
    """Write the prompt's K/V rows into the paged cache (in place) and return causal
    self-attention over the prompt, [seq_len, num_q_heads, head_dim]
    (01_paged_attention/paged_attention.py:104-136)."""
    seq_len = query.shape[0]
    table = _as_int_list(block_table)
    slots = compute_slot_mapping(table, seq_len, block_size)
    dev = _compute_device(query, kv_cache)
    q = query.to(dev)
    k = key.to(device=dev, dtype=kv_cache.dtype)
    v = value.to(device=dev, dtype=kv_cache.dtype)
    slot_t = _upload(slots, torch.int64, dev)
    if kv_cache.is_cuda:
        ops.kv_write_slots(kv_cache[0], kv_cache[1], k, v, slot_t, 'NHSD')
    else:
        staged = kv_cache.to(dev)
        ops.kv_write_slots(staged[0], staged[1], k, v, slot_t, 'NHSD')
        kv_cache.copy_(staged)
    cu = _upload([0, seq_len], torch.int32, dev)
    out = ops.prefill_causal(q, key.to(device=dev, dtype=q.dtype), value.to(device=dev, dtype=q.dtype),
                             cu, seq_len)
    return out.to(query.device)
