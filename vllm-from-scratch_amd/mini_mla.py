"""Drop-in `mini_mla` module: the reference's Multi-head Latent Attention names and signatures
(04_mla/mini_mla.py:25-362) with the attention core on MI355X matrix cores.

    rotate_half, apply_rotary_emb, RotaryEmbedding, MLAKVCache,
    MultiHeadLatentAttention, compare_kv_cache_size

plus, with no counterpart in the reference, a block cache for the latent rows (PagedMLAKVCache)
and MultiHeadLatentAttention.forward_paged, which attends through block tables
(mi355_mla_decode_paged).

The projections, norms and RoPE of the *query* are torch ops.  The attention itself never
expands the cached latent rows to per-head K/V as the reference does
(04_mla/mini_mla.py:245-248): the key up-projection is folded into the query and the value
up-projection is applied after attention (two small dense products), and the heads attend
directly to the cached rows [c_kv | k_rope] in libmi355_paged.so (mi355_mla_decode), which
also applies RoPE to the cached k_rope at read time.  The result equals the reference's up to
floating-point rounding; the core runs with 16-bit inputs (fp16 for fp32/fp16 modules, bf16
for bf16 modules) and fp32 accumulation.

Gradients: the forward pass always runs the kernel; when autograd needs a backward pass the
attention core is recomputed with differentiable torch ops inside backward() only.
Host tensors are staged through HIP device 0; there is no CPU implementation of the core.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from mi355_paged import ops


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def apply_rotary_emb(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor,
                     sin: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """RoPE on q and k [..., S, head_dim] with cos/sin [S, head_dim] (04_mla/mini_mla.py:32-44)."""
    cos, sin = cos[None, None], sin[None, None]
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


class RotaryEmbedding(nn.Module):
    """cos/sin tables [seq_len, dim]; forward(seq_len) returns the first seq_len rows and grows
    the table when asked for more (04_mla/mini_mla.py:47-72)."""

    def __init__(self, dim: int, max_seq_len: int = 8192, base: float = 10000.0):
        super().__init__()
        self.dim = dim
        self.max_seq_len = max_seq_len
        self.register_buffer('inv_freq', 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)))
        self._build_cache(max_seq_len)

    def _build_cache(self, seq_len: int):
        angles = torch.outer(torch.arange(seq_len, device=self.inv_freq.device).float(), self.inv_freq)
        table = torch.cat((angles, angles), dim=-1)
        self.register_buffer('cos_cached', table.cos())
        self.register_buffer('sin_cached', table.sin())
        self.max_seq_len = max(self.max_seq_len, seq_len)

    def forward(self, seq_len: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if seq_len > self.cos_cached.shape[0]:
            self._build_cache(seq_len)
        return self.cos_cached[:seq_len], self.sin_cached[:seq_len]


class MLAKVCache:
    """Dense per-sequence cache of the latent rows: [layers, batch, seq, kv_lora_rank + rope_dim]
    (04_mla/mini_mla.py:79-132)."""

    def __init__(self, max_batch_size: int, max_seq_len: int, num_layers: int, kv_lora_rank: int,
                 qk_rope_head_dim: int, device: torch.device, dtype: torch.dtype = torch.float16):
        self.kv_lora_rank = kv_lora_rank
        self.qk_rope_head_dim = qk_rope_head_dim
        self.cache = torch.zeros(num_layers, max_batch_size, max_seq_len, kv_lora_rank + qk_rope_head_dim,
                                 device=device, dtype=dtype)
        self.seq_lens = torch.zeros(max_batch_size, dtype=torch.long, device=device)

    def get(self, layer_id: int, batch_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        filled = int(self.seq_lens[:batch_size].max().item())
        rows = self.cache[layer_id, :batch_size, :filled]
        return rows[..., :self.kv_lora_rank], rows[..., self.kv_lora_rank:]

    def update(self, layer_id: int, c_kv: torch.Tensor, k_rope: torch.Tensor, seq_lens: torch.Tensor):
        batch, new_tokens, _ = c_kv.shape
        for b in range(batch):
            first = int(self.seq_lens[b].item())
            dst = self.cache[layer_id, b, first:first + new_tokens]
            dst[:, :self.kv_lora_rank] = c_kv[b]
            dst[:, self.kv_lora_rank:] = k_rope[b]
        self.seq_lens[:batch] += new_tokens


class PagedMLAKVCache:
    """Block cache of the latent rows: pools [layers, num_blocks, block_size, kv_lora_rank] and
    [layers, num_blocks, block_size, rope_dim] under one block id space (a pool each rather than
    c_kv | k_rope side by side in one row: rows of 1024 contiguous bytes at a 1024-byte stride
    measure 9 % faster through the decode kernel than the same rows at a 1152-byte stride), and one
    block table per sequence, with the blocks handed out by the engine's BlockAllocator
    (mini_vllm.BlockAllocator: the same ids, free list and hash map the K/V caches of
    paged_attention use), so that latent caches share the pool management -- and prefix sharing --
    of the paged K/V path.  The reference keeps dense per-sequence tensors (MLAKVCache above,
    04_mla/mini_mla.py:79-132); this class has no counterpart there.  block_size: a power of two."""

    def __init__(self, num_blocks: int, block_size: int, num_layers: int, kv_lora_rank: int,
                 qk_rope_head_dim: int, device, dtype: torch.dtype = torch.bfloat16, allocator=None):
        if block_size <= 0 or block_size & (block_size - 1):
            raise ValueError(f"block_size {block_size} is not a power of two")
        if allocator is None:
            from mini_vllm import BlockAllocator
            allocator = BlockAllocator(num_blocks, block_size)
        self.allocator = allocator
        self.block_size = block_size
        self.kv_lora_rank = kv_lora_rank
        self.qk_rope_head_dim = qk_rope_head_dim
        self.pool = torch.zeros(num_layers, num_blocks, block_size, kv_lora_rank, device=device, dtype=dtype)
        self.rope_pool = torch.zeros(num_layers, num_blocks, block_size, qk_rope_head_dim, device=device, dtype=dtype)
        self.block_tables: dict = {}     # sequence id -> list of block ids
        self.seq_lens: dict = {}         # sequence id -> tokens held (of the layer written last)
        self._layer_lens: dict = {}      # (sequence id, layer) -> tokens held

    def _slots(self, seq_id, first: int, count: int) -> torch.Tensor:
        table = self.block_tables.setdefault(seq_id, [])
        bs = self.block_size
        while len(table) * bs < first + count:
            blk = self.allocator.allocate()
            if blk is None:
                raise RuntimeError(f"OOM: no free latent-cache block for sequence {seq_id!r}")
            table.append(blk)
        pos = torch.arange(first, first + count)
        return torch.tensor(table, dtype=torch.long)[pos // bs] * bs + pos % bs

    def append(self, seq_id, layer_id: int, c_kv: torch.Tensor, k_rope: torch.Tensor) -> int:
        """Stores T new rows of one sequence ([T, kv_lora_rank], [T, rope_dim]) behind the rows the
        layer already holds; returns the sequence's length in that layer afterwards."""
        first = self._layer_lens.get((seq_id, layer_id), 0)
        count = c_kv.shape[0]
        slots = self._slots(seq_id, first, count).to(self.pool.device)
        self.pool[layer_id].view(-1, self.kv_lora_rank)[slots] = c_kv.to(device=self.pool.device, dtype=self.pool.dtype)
        self.rope_pool[layer_id].view(-1, self.qk_rope_head_dim)[slots] = k_rope.to(device=self.pool.device, dtype=self.pool.dtype)
        self._layer_lens[(seq_id, layer_id)] = first + count
        self.seq_lens[seq_id] = first + count
        return first + count

    def length(self, seq_id, layer_id: int) -> int:
        return self._layer_lens.get((seq_id, layer_id), 0)

    def tables(self, seq_ids) -> torch.Tensor:
        """int32 [len(seq_ids), max_blocks] on the pool's device, rows padded with block 0."""
        width = max(1, max(len(self.block_tables.get(s, ())) for s in seq_ids))
        rows = [list(self.block_tables.get(s, ())) + [0] * (width - len(self.block_tables.get(s, ()))) for s in seq_ids]
        return torch.tensor(rows, dtype=torch.int32).to(self.pool.device)

    def views(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(c_kv [blocks, block_size, R], k_rope [blocks, block_size, P]) of one layer."""
        return self.pool[layer_id], self.rope_pool[layer_id]

    def free(self, seq_id) -> None:
        for blk in self.block_tables.pop(seq_id, ()):
            self.allocator.free(blk)
        self.seq_lens.pop(seq_id, None)
        for key in [k for k in self._layer_lens if k[0] == seq_id]:
            del self._layer_lens[key]


def _compute_device(t: torch.Tensor) -> torch.device:
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("mini_mla: a HIP device is required for the attention core (no CPU implementation)")
    return torch.device('cuda', 0)


def _core_reference_for_backward(q_cat, c_kv, k_rope, cos, sin, context, row_seq, scale):
    """Differentiable restatement of the kernel's math; used ONLY to build gradients."""
    lora = c_kv.shape[-1]
    outs = []
    for r in range(q_cat.shape[0]):
        n, s = int(context[r]), int(row_seq[r])
        keys = k_rope[s, :n] * cos[:n] + rotate_half(k_rope[s, :n]) * sin[:n]
        scores = (q_cat[r, :, :lora] @ c_kv[s, :n].T + q_cat[r, :, lora:] @ keys.T) * scale
        outs.append(torch.softmax(scores, dim=-1) @ c_kv[s, :n])
    return torch.stack(outs)


class _LatentAttention(torch.autograd.Function):
    """forward = mi355_mla_decode; backward = autograd through a torch recomputation."""

    @staticmethod
    def forward(ctx, q_cat, c_kv, k_rope, cos, sin, context, row_seq, max_context, scale):
        dev = _compute_device(q_cat)
        # fp32 modules keep fp32 operands (fp32 matrix instruction: exact products, fp32 weights) where
        # that kernel serves the shape; otherwise, and for 16-bit modules, the 16-bit kernels run --
        # bf16 for everything but fp16 modules: un-normalised projections can exceed fp16's range
        if q_cat.dtype == torch.float32 and ops.mla_fp32_supported(c_kv.shape[-1], k_rope.shape[-1]):
            kdtype = torch.float32
        else:
            kdtype = torch.float16 if q_cat.dtype == torch.float16 else torch.bfloat16
        out = ops.mla_decode(q_cat.detach().to(device=dev, dtype=kdtype), c_kv.detach().to(device=dev, dtype=kdtype),
                             k_rope.detach().to(device=dev, dtype=kdtype), cos.to(dev), sin.to(dev),
                             context.to(device=dev, dtype=torch.int32), max_context, scale,
                             row_seq=row_seq.to(device=dev, dtype=torch.int32))
        ctx.save_for_backward(q_cat, c_kv, k_rope, cos, sin, context, row_seq)
        ctx.scale = scale
        return out.to(device=q_cat.device, dtype=q_cat.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        q_cat, c_kv, k_rope, cos, sin, context, row_seq = ctx.saved_tensors
        with torch.enable_grad():
            q = q_cat.detach().requires_grad_(True)
            c = c_kv.detach().requires_grad_(True)
            k = k_rope.detach().requires_grad_(True)
            out = _core_reference_for_backward(q, c, k, cos, sin, context, row_seq, ctx.scale)
            gq, gc, gk = torch.autograd.grad(out, (q, c, k), grad_out.to(out.dtype))
        return gq, gc, gk, None, None, None, None, None, None


class MultiHeadLatentAttention(nn.Module):
    """MLA layer: low-rank KV compression with decoupled RoPE (04_mla/mini_mla.py:139-306).
    Same constructor, parameter names and forward signature as the reference."""

    def __init__(self, hidden_size: int = 512, num_heads: int = 8, qk_nope_head_dim: int = 64,
                 qk_rope_head_dim: int = 32, kv_lora_rank: int = 128, v_head_dim: int = 64,
                 max_seq_len: int = 4096):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.qk_nope_head_dim = qk_nope_head_dim
        self.qk_rope_head_dim = qk_rope_head_dim
        self.kv_lora_rank = kv_lora_rank
        self.v_head_dim = v_head_dim
        q_head_dim = qk_nope_head_dim + qk_rope_head_dim
        q_lora_rank = num_heads * q_head_dim // 4
        self.q_a_proj = nn.Linear(hidden_size, q_lora_rank, bias=False)
        self.q_a_layernorm = nn.RMSNorm(q_lora_rank)
        self.q_b_proj = nn.Linear(q_lora_rank, num_heads * q_head_dim, bias=False)
        self.kv_a_proj_with_mqa = nn.Linear(hidden_size, kv_lora_rank + qk_rope_head_dim, bias=False)
        self.kv_a_layernorm = nn.RMSNorm(kv_lora_rank)
        self.kv_b_proj = nn.Linear(kv_lora_rank, num_heads * (qk_nope_head_dim + v_head_dim), bias=False)
        self.o_proj = nn.Linear(num_heads * v_head_dim, hidden_size, bias=False)
        self.rotary_emb = RotaryEmbedding(dim=qk_rope_head_dim, max_seq_len=max_seq_len)
        self.scale = q_head_dim ** -0.5

    def forward(self, hidden_states: torch.Tensor, positions: torch.Tensor,
                past_c_kv: Optional[torch.Tensor] = None, past_k_rope: Optional[torch.Tensor] = None):
        """hidden_states [B, T, hidden]; positions [B, T]; optional cached c_kv [B, past, lora] and
        k_rope [B, past, rope] -> (output [B, T, hidden], new_c_kv [B, T, lora], new_k_rope [B, T, rope])."""
        B, T, _ = hidden_states.shape
        H, nope, rope, lora, vd = (self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim,
                                   self.kv_lora_rank, self.v_head_dim)
        q = self.q_b_proj(self.q_a_layernorm(self.q_a_proj(hidden_states))).view(B, T, H, nope + rope)
        compressed = self.kv_a_proj_with_mqa(hidden_states)
        new_c_kv = self.kv_a_layernorm(compressed[..., :lora])
        new_k_rope = compressed[..., lora:]
        if past_c_kv is not None:
            c_kv = torch.cat((past_c_kv, new_c_kv), dim=1)
            k_rope = torch.cat((past_k_rope, new_k_rope), dim=1)
        else:
            c_kv, k_rope = new_c_kv, new_k_rope
        S = c_kv.shape[1]

        cos, sin = self.rotary_emb(int(max(int(positions.max().item()) + 1, S)))
        flat = positions.reshape(-1)
        cos_q = cos[flat].view(B, T, 1, rope)
        sin_q = sin[flat].view(B, T, 1, rope)
        q_rope = q[..., nope:] * cos_q + rotate_half(q[..., nope:]) * sin_q

        # fold the key up-projection into the query: q_lat[h] = W_k[h]^T q_nope[h]
        w_kv = self.kv_b_proj.weight.view(H, nope + vd, lora)
        q_lat = torch.einsum('bthn,hnl->bthl', q[..., :nope], w_kv[:, :nope])
        q_cat = torch.cat((q_lat, q_rope), dim=-1).reshape(B * T, H, lora + rope)

        # query token i of a T-token call sees cached tokens 0 .. S-T+i (the reference's mask diagonal)
        context = (torch.arange(T, device=positions.device) + (S - T + 1)).repeat(B)
        row_seq = torch.arange(B, device=positions.device).repeat_interleave(T)
        o_lat = _LatentAttention.apply(q_cat, c_kv, k_rope, cos[:S], sin[:S], context, row_seq, S, self.scale)

        out = torch.einsum('bthl,hvl->bthv', o_lat.view(B, T, H, lora), w_kv[:, nope:])
        return self.o_proj(out.reshape(B, T, H * vd)), new_c_kv, new_k_rope


    @torch.no_grad()
    def forward_paged(self, hidden_states: torch.Tensor, positions: torch.Tensor, cache: PagedMLAKVCache,
                      layer_id: int, seq_ids) -> torch.Tensor:
        """forward() over a block cache (inference only): the T new latent rows of every sequence
        in `seq_ids` (one per batch row) are appended to `cache`, and the heads attend to the
        sequence's cached rows through its block table (mi355_mla_decode_paged) -- no dense
        past_c_kv / past_k_rope tensors are built.  hidden_states [B, T, hidden], positions [B, T]
        -> output [B, T, hidden]; equals forward() fed with the same rows as dense tensors."""
        B, T, _ = hidden_states.shape
        H, nope, rope, lora, vd = (self.num_heads, self.qk_nope_head_dim, self.qk_rope_head_dim,
                                   self.kv_lora_rank, self.v_head_dim)
        q = self.q_b_proj(self.q_a_layernorm(self.q_a_proj(hidden_states))).view(B, T, H, nope + rope)
        compressed = self.kv_a_proj_with_mqa(hidden_states)
        new_c_kv = self.kv_a_layernorm(compressed[..., :lora])
        new_k_rope = compressed[..., lora:]
        lens = [cache.append(seq_ids[b], layer_id, new_c_kv[b], new_k_rope[b]) for b in range(B)]
        S = max(lens)
        cos, sin = self.rotary_emb(int(max(int(positions.max().item()) + 1, S)))
        flat = positions.reshape(-1)
        q_rope = q[..., nope:] * cos[flat].view(B, T, 1, rope) + rotate_half(q[..., nope:]) * sin[flat].view(B, T, 1, rope)
        w_kv = self.kv_b_proj.weight.view(H, nope + vd, lora)
        q_lat = torch.einsum('bthn,hnl->bthl', q[..., :nope], w_kv[:, :nope])
        q_cat = torch.cat((q_lat, q_rope), dim=-1).reshape(B * T, H, lora + rope)
        dev = cache.pool.device
        kdtype = cache.pool.dtype
        # query token i of sequence b sees cached tokens 0 .. len_b - T + i
        context = torch.tensor([n - T + i + 1 for n in lens for i in range(T)], dtype=torch.int32, device=dev)
        row_seq = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(T)
        c_view, r_view = cache.views(layer_id)
        o_lat = ops.mla_decode_paged(q_cat.to(device=dev, dtype=kdtype), c_view, r_view, cache.tables(seq_ids),
                                     cos[:S].to(dev), sin[:S].to(dev), context, S, self.scale, row_seq=row_seq)
        o_lat = o_lat.to(device=hidden_states.device, dtype=hidden_states.dtype)
        out = torch.einsum('bthl,hvl->bthv', o_lat.view(B, T, H, lora), w_kv[:, nope:])
        return self.o_proj(out.reshape(B, T, H * vd))


def compare_kv_cache_size(num_heads: int = 32, head_dim: int = 128, kv_lora_rank: int = 512,
                          qk_rope_head_dim: int = 64, num_layers: int = 27, seq_len: int = 4096,
                          batch_size: int = 8, dtype_bytes: int = 2):
    """KV-cache megabytes of standard multi-head attention vs MLA -> (mha_mb, mla_mb)
    (04_mla/mini_mla.py:313-362)."""
    per_token = {'MHA': num_heads * head_dim * 2, 'MLA': kv_lora_rank + qk_rope_head_dim}
    tokens = batch_size * seq_len * num_layers
    mha_mb, mla_mb = (tokens * per_token[k] * dtype_bytes / 1e6 for k in ('MHA', 'MLA'))
    print(f"KV cache, {num_layers} layers x batch {batch_size} x {seq_len} tokens, {dtype_bytes} B/element")
    print(f"  MHA: {per_token['MHA']} elements per token per layer ({num_heads} heads x {head_dim} x K,V) -> {mha_mb:.1f} MB")
    print(f"  MLA: {per_token['MLA']} elements per token per layer (c_kv {kv_lora_rank} + k_rope {qk_rope_head_dim}) "
          f"-> {mla_mb:.1f} MB  ({mha_mb / mla_mb:.1f}x smaller)")
    return mha_mb, mla_mb
