"""Tensor-level wrappers over the C ABI: torch tensors in, torch tensors out.

torch is used for device memory, streams and dtype bookkeeping only; every
operation here is one call into libmi355_paged.so on the tensor's HIP device.
Tensors must already live on the device -- the drop-in modules
(paged_attention.py / mini_vllm.py) do the staging for host tensors.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib

_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
SUPPORTED_HEAD_DIMS = (16, 32, 64, 128, 256)

# KV cache layouts, as (block, head, token) dimension indices of a 4-D cache tensor:
#   'NHSD' = [num_blocks, kv_heads, block_size, head_dim]   (KVCachePool.kv_cache[0|1, layer])
#   'NSHD' = [num_blocks, block_size, heads, head_dim]      (MiniTransformer.k_cache[layer])
_LAYOUT_DIMS = {'NHSD': (0, 1, 2), 'NSHD': (0, 2, 1)}

_workspaces = {}

_KNOB_RESET = -2 ** 31


class debug_knob:
    """`with ops.debug_knob('MI355_PAGED_HPW', 1): ...` -- one launch-plan knob of the library set
    for the block (mi355_debug_set_knob), then back to the process's start-up value.  For A/B
    measurements and the tests that compare two kernels on the same inputs."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        _lib.check(_lib.load().mi355_debug_set_knob(self.name, self.value), 'mi355_debug_set_knob')
        return self

    def __exit__(self, *exc):
        _lib.load().mi355_debug_set_knob(self.name, _KNOB_RESET)
        return False


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise TypeError(f"mi355_paged: unsupported dtype {t.dtype}; use float32, bfloat16 or float16")


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mi355_paged: tensors must live on a HIP device (torch device 'cuda'); "
                               "this path has no CPU implementation")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"mi355_paged: tensors on different devices ({dev} vs {t.device})")
    return dev


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(dev: torch.device) -> int:
    """Handle of torch's current stream on `dev` (the raw getter skips building a Stream object:
    a fraction of a microsecond instead of several, on a path that runs once per kernel launch)."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev: torch.device):
    """Context in which `dev` is the current HIP device; free when it already is (the usual
    case: torch's device guard costs several microseconds per call)."""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NO_SWITCH
    return torch.cuda.device(dev)


def _workspace(dev: torch.device, nbytes: int, stream: int) -> Tuple[Optional[torch.Tensor], int]:
    """A scratch tensor per (device, stream) that only ever grows (torch owns the memory): two
    streams running split launches at the same time must not share partials.

    While a HIP graph is being captured the scratch comes from the capture's own memory pool
    instead: a captured launch keeps the raw pointer, and the shared tensor may be replaced by a
    larger one (and freed) long before the graph is replayed."""
    if nbytes == 0:
        return None, 0
    if torch.cuda.is_current_stream_capturing():
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws, ws.numel()
    key = (dev, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws, ws.numel()


def _stride(t: torch.Tensor, dim: int) -> int:
    """Stride of `dim`; a size-1 dim may carry any stride, so report the contiguous one."""
    if t.shape[dim] != 1:
        return t.stride(dim)
    n = 1
    for d in range(dim + 1, t.dim()):
        n *= t.shape[d]
    return n


def _checked_out(out: Optional[torch.Tensor], shape, dtype, dev) -> torch.Tensor:
    """A caller-supplied result tensor must be exactly what the kernel writes: the kernels take a
    bare pointer and assume a dense tensor of this shape, so anything else would be silently
    written in the wrong places."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {dev}, got "
                         f"{out.dtype} {tuple(out.shape)} strides {out.stride()} on {out.device}")
    return out


def _kv_geometry(cache: torch.Tensor, layout: str):
    if cache.dim() != 4:
        raise ValueError(f"KV cache must be 4-D ({layout}), got shape {tuple(cache.shape)}")
    if cache.stride(3) != 1:
        raise ValueError("KV cache head rows must be contiguous (stride of the last dim == 1)")
    bi, hi, ti = _LAYOUT_DIMS[layout]
    return dict(num_blocks=cache.shape[bi], heads=cache.shape[hi], block_size=cache.shape[ti],
                head_dim=cache.shape[3], s_block=_stride(cache, bi), s_head=_stride(cache, hi),
                s_tok=_stride(cache, ti))


def paged_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                 block_tables: torch.Tensor, context_lens: torch.Tensor, max_context: int,
                 layout: str = 'NHSD', scale: Optional[float] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched paged decode attention.

    q [B, Hq, D]; k_cache / v_cache 4-D in `layout`; block_tables [B, max_blocks] int32;
    context_lens [B] int32; max_context = host-side upper bound of context_lens.
    Returns out [B, Hq, D] in q's dtype.
    """
    dev = require_device(q, k_cache, v_cache, block_tables, context_lens)
    lib = _lib.load()
    if q.dim() != 3 or q.stride(2) != 1 or q.stride(1) != q.shape[2]:
        q = q.contiguous()
    B, Hq, D = q.shape
    g = _kv_geometry(k_cache, layout)
    if tuple(v_cache.shape) != tuple(k_cache.shape) or _kv_geometry(v_cache, layout) != g:
        raise ValueError("k_cache and v_cache must share shape and strides")
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise TypeError("query and KV cache dtypes differ")
    if g['head_dim'] != D:
        raise ValueError(f"head_dim mismatch: q {D}, cache {g['head_dim']}")
    if block_tables.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise TypeError("block_tables and context_lens must be int32")
    block_tables = block_tables.contiguous()
    context_lens = context_lens.contiguous()
    if block_tables.dim() != 2 or block_tables.shape[0] != B or context_lens.numel() != B:
        raise ValueError("block_tables must be [B, max_blocks] and context_lens [B]")
    out = _checked_out(out, (B, Hq, D), q.dtype, dev)
    code = dtype_code(q)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    # the library plans for the context a table row can hold; size the workspace for that same bound
    max_context = min(int(max_context), block_tables.shape[1] * g['block_size'])
    with _on(dev):
        stream = _stream(dev)
        need = lib.mi355_paged_decode_workspace_bytes(B, Hq, g['heads'], D, g['block_size'],
                                                      max_context, code)
        ws, ws_bytes = _workspace(dev, need, stream)
        rc = lib.mi355_paged_decode(
            out.data_ptr(), q.data_ptr(), _stride(q, 0), k_cache.data_ptr(), v_cache.data_ptr(),
            block_tables.data_ptr(), context_lens.data_ptr(), B, Hq, g['heads'], D,
            g['block_size'], block_tables.shape[1], g['num_blocks'], max_context,
            g['s_block'], g['s_head'], g['s_tok'], float(scale), code,
            ws.data_ptr() if ws is not None else None, ws_bytes, stream)
    _lib.check(rc, 'mi355_paged_decode')
    return out


def paged_decode_append(q: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor, k_cache: torch.Tensor,
                        v_cache: torch.Tensor, block_tables: torch.Tensor, context_lens: torch.Tensor,
                        max_context: int, layout: str = 'NHSD', scale: Optional[float] = None,
                        out: Optional[torch.Tensor] = None,
                        rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """One decode step of an attention layer: k_new / v_new [B, Hkv, D] go to position
    context_lens[b] - 1 of every sequence's cache, then q attends to the context_lens[b] cached
    rows.  Same result (output and caches) as kv_write_slots at those slots followed by
    paged_decode; where the VALU decode kernel runs, in one launch.
    rope = (cos_table, sin_table), each [table_len, D] in the cache's dtype: q and k_new are given
    UN-rotated and are rotated for position context_lens[b] - 1 inside the launch (q itself is left
    as it is); raises Mi355Error (status MI355_EUNSUPPORTED, nothing launched) for the shapes whose
    cache write is a launch of its own -- see paged_decode_append_rope_serves."""
    dev = require_device(q, k_new, v_new, k_cache, v_cache, block_tables, context_lens)
    lib = _lib.load()
    if q.dim() != 3 or q.stride(2) != 1 or q.stride(1) != q.shape[2]:
        q = q.contiguous()
    B, Hq, D = q.shape
    g = _kv_geometry(k_cache, layout)
    if tuple(v_cache.shape) != tuple(k_cache.shape) or _kv_geometry(v_cache, layout) != g:
        raise ValueError("k_cache and v_cache must share shape and strides")
    if not (k_cache.dtype == v_cache.dtype == k_new.dtype == v_new.dtype == q.dtype):
        raise TypeError("query, new rows and KV cache dtypes differ")
    if g['head_dim'] != D:
        raise ValueError(f"head_dim mismatch: q {D}, cache {g['head_dim']}")

    def rows(t):
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            t = t.contiguous()
        return t

    k_new, v_new = rows(k_new), rows(v_new)
    if _stride(v_new, 0) != _stride(k_new, 0):
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
    if tuple(k_new.shape) != (B, g['heads'], D) or tuple(v_new.shape) != (B, g['heads'], D):
        raise ValueError("k_new / v_new must be [B, kv_heads, head_dim] matching the cache")
    if block_tables.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise TypeError("block_tables and context_lens must be int32")
    block_tables = block_tables.contiguous()
    context_lens = context_lens.contiguous()
    if block_tables.dim() != 2 or block_tables.shape[0] != B or context_lens.numel() != B:
        raise ValueError("block_tables must be [B, max_blocks] and context_lens [B]")
    out = _checked_out(out, (B, Hq, D), q.dtype, dev)
    code = dtype_code(q)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    max_context = min(int(max_context), block_tables.shape[1] * g['block_size'])
    with _on(dev):
        stream = _stream(dev)
        need = lib.mi355_paged_decode_workspace_bytes(B, Hq, g['heads'], D, g['block_size'],
                                                      max_context, code)
        ws, ws_bytes = _workspace(dev, need, stream)
        if rope is None:
            rc = lib.mi355_paged_decode_append(
                out.data_ptr(), q.data_ptr(), _stride(q, 0), k_new.data_ptr(), v_new.data_ptr(), _stride(k_new, 0),
                k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(), context_lens.data_ptr(),
                B, Hq, g['heads'], D, g['block_size'], block_tables.shape[1], g['num_blocks'], max_context,
                g['s_block'], g['s_head'], g['s_tok'], float(scale), code,
                ws.data_ptr() if ws is not None else None, ws_bytes, stream)
        else:
            cos_t, sin_t = rope
            if not (cos_t.dtype == sin_t.dtype == q.dtype) or cos_t.shape != sin_t.shape or cos_t.dim() != 2 \
                    or cos_t.shape[1] != D or not cos_t.is_contiguous() or not sin_t.is_contiguous() \
                    or cos_t.device != dev or sin_t.device != dev:
                raise ValueError("rope tables must be contiguous [table_len, head_dim] tensors of q's dtype and device")
            rc = lib.mi355_paged_decode_append_rope(
                out.data_ptr(), q.data_ptr(), _stride(q, 0), k_new.data_ptr(), v_new.data_ptr(), _stride(k_new, 0),
                k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr(), context_lens.data_ptr(),
                cos_t.data_ptr(), sin_t.data_ptr(), cos_t.shape[0],
                B, Hq, g['heads'], D, g['block_size'], block_tables.shape[1], g['num_blocks'], max_context,
                g['s_block'], g['s_head'], g['s_tok'], float(scale), code,
                ws.data_ptr() if ws is not None else None, ws_bytes, stream)
    _lib.check(rc, 'mi355_paged_decode_append' if rope is None else 'mi355_paged_decode_append_rope')
    return out


def paged_decode_append_rope_serves(dtype: torch.dtype, heads_q: int, heads_kv: int, head_dim: int) -> bool:
    """Whether paged_decode_append(rope=...) runs as one launch for this geometry: everything the VALU
    decode kernel serves, i.e. not the grouped-query shapes of the matrix-core kernel (16-bit,
    head_dim 64 / 128, four or more query heads per kv head).  (A plan whose LDS is full is refused
    at the call; that takes contexts beyond what a workgroup stages.)"""
    if head_dim not in (16, 32, 64, 128, 256) or heads_kv <= 0 or heads_q % heads_kv:
        return False
    grouped = heads_q // heads_kv >= 4 and dtype in (torch.bfloat16, torch.float16) and head_dim in (64, 128)
    return not grouped


def kv_write_slots(k_cache: torch.Tensor, v_cache: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                   slots: torch.Tensor, layout: str = 'NHSD') -> None:
    """cache[slot // BS, :, slot % BS, :] = row, for T rows; k, v [T, Hkv, D]; slots [T] int64."""
    dev = require_device(k_cache, v_cache, k, v, slots)
    lib = _lib.load()
    g = _kv_geometry(k_cache, layout)
    if tuple(v_cache.shape) != tuple(k_cache.shape) or _kv_geometry(v_cache, layout) != g:
        raise ValueError("k_cache and v_cache must share shape and strides")
    if k.dtype != k_cache.dtype or v.dtype != k_cache.dtype:
        raise TypeError("new K/V rows and the cache differ in dtype")

    def rows(t):
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            t = t.contiguous()
        return t

    k, v = rows(k), rows(v)
    if _stride(v, 0) != _stride(k, 0):
        k, v = k.contiguous(), v.contiguous()
    T, H, D = k.shape
    if H != g['heads'] or D != g['head_dim'] or tuple(v.shape) != (T, H, D):
        raise ValueError("k/v must be [T, kv_heads, head_dim] matching the cache")
    if slots.dtype != torch.int64:
        slots = slots.to(torch.int64)
    slots = slots.contiguous()
    if slots.numel() != T:
        raise ValueError("one slot per token expected")
    with _on(dev):
        rc = lib.mi355_kv_write_slots(
            k_cache.data_ptr(), v_cache.data_ptr(), k.data_ptr(), v.data_ptr(), slots.data_ptr(),
            T, H, D, _stride(k, 0), g['block_size'], g['num_blocks'], g['s_block'], g['s_head'],
            g['s_tok'], dtype_code(k), _stream(dev))
    _lib.check(rc, 'mi355_kv_write_slots')


def prefill_causal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens: torch.Tensor,
                   max_seqlen: int, scale: Optional[float] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal attention over packed sequences. q [T, Hq, D]; k, v [T, Hkv, D];
    cu_seqlens [nseq + 1] int32 (device). Returns [T, Hq, D] (written into `out` when given)."""
    dev = require_device(q, k, v, cu_seqlens)
    lib = _lib.load()

    def rows(t):
        if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            t = t.contiguous()
        return t

    q, k, v = rows(q), rows(k), rows(v)
    if _stride(v, 0) != _stride(k, 0):
        k, v = k.contiguous(), v.contiguous()
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    if k.shape[0] != T or tuple(v.shape) != tuple(k.shape) or k.shape[2] != D:
        raise ValueError("q/k/v token counts or head_dim differ")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError("q/k/v dtypes differ")
    if cu_seqlens.dtype != torch.int32:
        raise TypeError("cu_seqlens must be int32")
    cu_seqlens = cu_seqlens.contiguous()
    out = _checked_out(out, (T, Hq, D), q.dtype, dev)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    with _on(dev):
        rc = lib.mi355_prefill_causal(
            out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), _stride(q, 0), _stride(k, 0),
            cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, T, int(max_seqlen), Hq, Hkv, D,
            float(scale), dtype_code(q), _stream(dev))
    _lib.check(rc, 'mi355_prefill_causal')
    return out


def sample_argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-wise argmax of [B, vocab] logits -> int32 [B] (ties: lowest index; NaN wins)."""
    dev = require_device(logits)
    lib = _lib.load()
    if logits.dim() != 2:
        raise ValueError("logits must be [batch, vocab]")
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    B, V = logits.shape
    if out is None:
        out = torch.empty((B,), dtype=torch.int32, device=dev)
    elif out.dtype != torch.int32 or out.numel() < B or not out.is_contiguous() or out.device != dev:
        raise ValueError("out must be a contiguous int32 vector of at least B entries on the logits' device")
    with _on(dev):
        rc = lib.mi355_sample_argmax(out.data_ptr(), logits.data_ptr(), B, V, _stride(logits, 0),
                                     dtype_code(logits), _stream(dev))
    _lib.check(rc, 'mi355_sample_argmax')
    return out


def sample_topk_topp(logits: torch.Tensor, temperature: torch.Tensor,
                     top_p: Optional[torch.Tensor] = None, top_k: Optional[torch.Tensor] = None,
                     seed: int = 0, offset: int = 0, return_probs: bool = False,
                     out: Optional[torch.Tensor] = None):
    """One draw per row after temperature / top-k / top-p filtering -> int32 [B]
    (and, with return_probs, the renormalised post-filter distribution [B, vocab] fp32)."""
    dev = require_device(logits, temperature, top_p, top_k)
    lib = _lib.load()
    if logits.dim() != 2:
        raise ValueError("logits must be [batch, vocab]")
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    B, V = logits.shape
    temperature = temperature.to(torch.float32).contiguous()
    if top_p is not None:
        top_p = top_p.to(torch.float32).contiguous()
    if top_k is not None:
        top_k = top_k.to(torch.int32).contiguous()
    if out is None:
        out = torch.empty((B,), dtype=torch.int32, device=dev)
    elif out.dtype != torch.int32 or out.numel() < B or not out.is_contiguous():
        raise ValueError("out must be a contiguous int32 vector of at least B entries")
    probs = torch.empty((B, V), dtype=torch.float32, device=dev) if return_probs else None
    with _on(dev):
        rc = lib.mi355_sample_topk_topp(
            out.data_ptr(), logits.data_ptr(), temperature.data_ptr(),
            top_p.data_ptr() if top_p is not None else None,
            top_k.data_ptr() if top_k is not None else None,
            int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), B, V, _stride(logits, 0),
            dtype_code(logits), probs.data_ptr() if probs is not None else None, _stream(dev))
    _lib.check(rc, 'mi355_sample_topk_topp')
    return (out, probs) if return_probs else out


def slot_mapping(block_tables: torch.Tensor, starts: torch.Tensor, cu_tokens: torch.Tensor,
                 total_tokens: int, block_size: int) -> torch.Tensor:
    """Device slot mapping: for sequence b, tokens at positions starts[b] + i land in
    out[cu_tokens[b] + i] = block_tables[b][pos // BS] * BS + pos % BS. Returns int64 [total]."""
    dev = require_device(block_tables, starts, cu_tokens)
    lib = _lib.load()
    for t in (block_tables, starts, cu_tokens):
        if t.dtype != torch.int32:
            raise TypeError("block_tables, starts and cu_tokens must be int32")
    block_tables = block_tables.contiguous()
    out = torch.empty((total_tokens,), dtype=torch.int64, device=dev)
    with _on(dev):
        rc = lib.mi355_slot_mapping(out.data_ptr(), block_tables.data_ptr(),
                                    starts.contiguous().data_ptr(), cu_tokens.contiguous().data_ptr(),
                                    block_tables.shape[0], block_tables.shape[1], int(block_size),
                                    int(total_tokens), _stream(dev))
    _lib.check(rc, 'mi355_slot_mapping')
    return out


def mla_fp32_supported(R: int, P: int) -> bool:
    """Shapes the fp32 MLA kernel serves (the 16-bit kernels take every R in {32..512}, P <= 64)."""
    return (R in (32, 64, 128) and P in (8, 16, 32, 64)) or (R in (256, 512) and P in (32, 64))


def _mla_out(out: Optional[torch.Tensor], rows: int, H: int, R: int, dev: torch.device) -> torch.Tensor:
    if out is None:
        return torch.empty((rows, H, R), dtype=torch.float32, device=dev)
    if out.dtype != torch.float32 or tuple(out.shape) != (rows, H, R) or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"out must be a contiguous fp32 [{rows}, {H}, {R}] tensor on {dev}")
    return out


def mla_decode(q: torch.Tensor, c_kv: torch.Tensor, k_rope: torch.Tensor, cos: torch.Tensor,
               sin: torch.Tensor, context_lens: torch.Tensor, max_context: int, scale: float,
               row_seq: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Latent-space MLA attention.

    q [rows, H, R + P] (absorbed nope query | rotated rope query); c_kv [S_seqs, S, R] and
    k_rope [S_seqs, S, P] (un-rotated) in q's dtype (bf16 / fp16, or fp32 where
    mla_fp32_supported(R, P)); cos / sin [table_len, P] fp32;
    context_lens [rows] int32; row_seq [rows] int32 or None.  Returns [rows, H, R] fp32
    (written into `out` when one is given: contiguous fp32 of that shape).
    """
    dev = require_device(q, c_kv, k_rope, cos, sin, context_lens, row_seq)
    lib = _lib.load()
    if q.dim() != 3 or c_kv.dim() != 3 or k_rope.dim() != 3:
        raise ValueError("q [rows,H,R+P], c_kv [seqs,S,R], k_rope [seqs,S,P] expected")
    rows, H, DQ = q.shape
    R, P = c_kv.shape[2], k_rope.shape[2]
    if DQ != R + P:
        raise ValueError(f"q last dim {DQ} != kv_lora_rank {R} + rope dim {P}")
    if c_kv.dtype != q.dtype or k_rope.dtype != q.dtype:
        raise TypeError("q, c_kv and k_rope must share one dtype")
    if q.stride(2) != 1 or q.stride(1) != DQ:
        q = q.contiguous()
    if c_kv.stride(2) != 1:
        c_kv = c_kv.contiguous()
    if k_rope.stride(2) != 1:
        k_rope = k_rope.contiguous()
    cos = cos.to(torch.float32).contiguous()
    sin = sin.to(torch.float32).contiguous()
    if cos.shape[-1] != P or tuple(sin.shape) != tuple(cos.shape):
        raise ValueError("cos / sin must be [table_len, rope_dim]")
    if context_lens.dtype != torch.int32 or (row_seq is not None and row_seq.dtype != torch.int32):
        raise TypeError("context_lens / row_seq must be int32")
    if tuple(k_rope.shape[:2]) != tuple(c_kv.shape[:2]):
        raise ValueError("c_kv and k_rope must hold the same [seqs, S] rows")
    if context_lens.numel() != rows or (row_seq is not None and row_seq.numel() != rows):
        raise ValueError("context_lens / row_seq must have one entry per query row")
    if row_seq is None and rows > c_kv.shape[0]:
        raise ValueError(f"{rows} query rows but {c_kv.shape[0]} cached sequences (pass row_seq to share them)")
    # the kernel clamps each row's context to max_context and reads that many cache rows and table rows
    if int(max_context) > c_kv.shape[1] or int(max_context) > cos.shape[0]:
        raise ValueError(f"max_context {max_context} exceeds the cache rows ({c_kv.shape[1]}) or the "
                         f"cos/sin table ({cos.shape[0]})")
    out = _mla_out(out, rows, H, R, dev)
    code = dtype_code(q)
    with _on(dev):
        stream = _stream(dev)
        need = lib.mi355_mla_decode_workspace_bytes(rows, H, R, P, int(max_context), code)
        ws, ws_bytes = _workspace(dev, need, stream)
        rc = lib.mi355_mla_decode(
            out.data_ptr(), q.data_ptr(), _stride(q, 0),
            c_kv.data_ptr(), _stride(c_kv, 0), _stride(c_kv, 1),
            k_rope.data_ptr(), _stride(k_rope, 0), _stride(k_rope, 1),
            cos.data_ptr(), sin.data_ptr(), cos.shape[0], context_lens.contiguous().data_ptr(),
            row_seq.contiguous().data_ptr() if row_seq is not None else None,
            rows, H, R, P, int(max_context), float(scale), code,
            ws.data_ptr() if ws is not None else None, ws_bytes, stream)
    _lib.check(rc, 'mi355_mla_decode')
    return out


def mla_decode_paged(q: torch.Tensor, c_kv: torch.Tensor, k_rope: torch.Tensor, block_tables: torch.Tensor,
                     cos: torch.Tensor, sin: torch.Tensor, context_lens: torch.Tensor, max_context: int, scale: float,
                     row_seq: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Latent-space MLA attention over a block cache (mi355_mla_decode_paged).

    c_kv [blocks, block_size, R] and k_rope [blocks, block_size, P] -- typically the two column
    ranges pool[..., :R] and pool[..., R:] of one pool [blocks, block_size, R + P] -- in q's dtype;
    block_tables [sequences, max_blocks] int32: token t of sequence s sits in block
    block_tables[s][t // block_size], row t % block_size.  Everything else as in mla_decode, whose
    result it equals bit for bit on the same rows.  block_size must be a power of two."""
    dev = require_device(q, c_kv, k_rope, block_tables, cos, sin, context_lens, row_seq)
    lib = _lib.load()
    if q.dim() != 3 or c_kv.dim() != 3 or k_rope.dim() != 3:
        raise ValueError("q [rows,H,R+P], c_kv [blocks,block_size,R], k_rope [blocks,block_size,P] expected")
    rows, H, DQ = q.shape
    R, P = c_kv.shape[2], k_rope.shape[2]
    if DQ != R + P:
        raise ValueError(f"q last dim {DQ} != kv_lora_rank {R} + rope dim {P}")
    if c_kv.dtype != q.dtype or k_rope.dtype != q.dtype:
        raise TypeError("q, c_kv and k_rope must share one dtype")
    if tuple(k_rope.shape[:2]) != tuple(c_kv.shape[:2]):
        raise ValueError("c_kv and k_rope must hold the same [blocks, block_size] rows")
    if q.stride(2) != 1 or q.stride(1) != DQ:
        q = q.contiguous()
    if c_kv.stride(2) != 1 or k_rope.stride(2) != 1:
        raise ValueError("cache rows must be contiguous in their last dimension (views of a pool are fine)")
    num_blocks, block_size = c_kv.shape[0], c_kv.shape[1]
    cos = cos.to(torch.float32).contiguous()
    sin = sin.to(torch.float32).contiguous()
    if cos.shape[-1] != P or tuple(sin.shape) != tuple(cos.shape):
        raise ValueError("cos / sin must be [table_len, rope_dim]")
    if block_tables.dtype != torch.int32 or context_lens.dtype != torch.int32 or \
            (row_seq is not None and row_seq.dtype != torch.int32):
        raise TypeError("block_tables / context_lens / row_seq must be int32")
    block_tables = block_tables.contiguous()
    if block_tables.dim() != 2:
        raise ValueError("block_tables must be [sequences, max_blocks]")
    if context_lens.numel() != rows or (row_seq is not None and row_seq.numel() != rows):
        raise ValueError("context_lens / row_seq must have one entry per query row")
    if row_seq is None and rows > block_tables.shape[0]:
        raise ValueError(f"{rows} query rows but {block_tables.shape[0]} block-table rows (pass row_seq to share them)")
    max_context = min(int(max_context), block_tables.shape[1] * block_size)
    if max_context > cos.shape[0]:
        raise ValueError(f"max_context {max_context} exceeds the cos/sin table ({cos.shape[0]})")
    out = _mla_out(out, rows, H, R, dev)
    code = dtype_code(q)
    with _on(dev):
        stream = _stream(dev)
        need = lib.mi355_mla_decode_workspace_bytes(rows, H, R, P, max_context, code)
        ws, ws_bytes = _workspace(dev, need, stream)
        rc = lib.mi355_mla_decode_paged(
            out.data_ptr(), q.data_ptr(), _stride(q, 0),
            c_kv.data_ptr(), _stride(c_kv, 0), _stride(c_kv, 1),
            k_rope.data_ptr(), _stride(k_rope, 0), _stride(k_rope, 1),
            block_tables.data_ptr(), block_size, block_tables.shape[1], num_blocks,
            cos.data_ptr(), sin.data_ptr(), cos.shape[0], context_lens.contiguous().data_ptr(),
            row_seq.contiguous().data_ptr() if row_seq is not None else None,
            rows, H, R, P, max_context, float(scale), code,
            ws.data_ptr() if ws is not None else None, ws_bytes, stream)
    _lib.check(rc, 'mi355_mla_decode_paged')
    return out


def _head_rows(t: torch.Tensor) -> torch.Tensor:
    """[T, H, D] with contiguous head rows packed D apart (any token stride)."""
    if t.dim() != 3 or t.stride(2) != 1 or (t.shape[1] != 1 and t.stride(1) != t.shape[2]):
        raise ValueError("expected [T, H, D] with heads packed D apart and contiguous head rows")
    return t


def rope_kv_write(q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor], cos_table: torch.Tensor,
                  sin_table: torch.Tensor, positions: torch.Tensor, slots: Optional[torch.Tensor] = None,
                  k_cache: Optional[torch.Tensor] = None, v_cache: Optional[torch.Tensor] = None,
                  layout: str = 'NSHD') -> None:
    """Rotate q [T, Hq, D] and k [T, Hkv, D] IN PLACE (rotate-half RoPE with the table rows
    picked by `positions`) and, when `slots` is given, write the rotated k and v [T, Hkv, D]
    into the paged caches at those slots.  q / k / v may be views into one projection output."""
    dev = require_device(q, k, v, cos_table, sin_table, positions, slots, k_cache, v_cache)
    lib = _lib.load()
    q, k = _head_rows(q), _head_rows(k)
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    if k.shape[0] != T or k.shape[2] != D or k.dtype != q.dtype:
        raise ValueError("q and k must agree in token count, head_dim and dtype")
    if cos_table.dtype != q.dtype or sin_table.dtype != q.dtype:
        raise TypeError("cos / sin tables must have the dtype of q")
    if cos_table.dim() != 2 or cos_table.shape[1] != D or tuple(sin_table.shape) != tuple(cos_table.shape) \
            or not cos_table.is_contiguous() or not sin_table.is_contiguous():
        raise ValueError("cos / sin tables must be contiguous [table_len, head_dim]")
    if positions.dtype != torch.int64 or positions.numel() != T:
        raise TypeError("positions must be int64 [T]")
    positions = positions.contiguous()
    geo = dict(block_size=0, num_blocks=0, s_block=0, s_head=0, s_tok=0)
    v_stride = 0
    if slots is not None:
        if v is None or k_cache is None or v_cache is None:
            raise ValueError("slots given without v / caches")
        v = _head_rows(v)
        if tuple(v.shape) != tuple(k.shape) or v.dtype != q.dtype or k_cache.dtype != q.dtype:
            raise ValueError("v must match k; caches must match the dtype of q")
        geo = _kv_geometry(k_cache, layout)
        if tuple(v_cache.shape) != tuple(k_cache.shape) or _kv_geometry(v_cache, layout) != geo:
            raise ValueError("k_cache and v_cache must share shape and strides")
        if geo['heads'] != Hkv or geo['head_dim'] != D:
            raise ValueError("cache heads / head_dim differ from k")
        if slots.dtype != torch.int64 or slots.numel() != T:
            raise TypeError("slots must be int64 [T]")
        slots = slots.contiguous()
        v_stride = _stride(v, 0)
    with _on(dev):
        rc = lib.mi355_rope_kv_write(
            q.data_ptr(), k.data_ptr(), v.data_ptr() if slots is not None else None,
            _stride(q, 0), _stride(k, 0), v_stride, cos_table.data_ptr(), sin_table.data_ptr(),
            cos_table.shape[0], positions.data_ptr(), slots.data_ptr() if slots is not None else None,
            k_cache.data_ptr() if slots is not None else None, v_cache.data_ptr() if slots is not None else None,
            T, Hq, Hkv, D, geo['block_size'], geo['num_blocks'], geo['s_block'], geo['s_head'], geo['s_tok'],
            dtype_code(q), _stream(dev))
    _lib.check(rc, 'mi355_rope_kv_write')


RMSNORM_MAX_HIDDEN = {torch.float32: 4096, torch.bfloat16: 8192, torch.float16: 8192}


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, delta: Optional[torch.Tensor] = None):
    """weight * (s / sqrt(mean(s^2) + eps)) per row of s = x (+ delta).  x, delta [T, hidden].
    Returns the normalised rows, or (x + delta, normalised rows) when delta is given."""
    dev = require_device(x, weight, delta)
    lib = _lib.load()
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be [T, hidden] with contiguous rows")
    T, hidden = x.shape
    if weight.dtype != x.dtype or weight.numel() != hidden or not weight.is_contiguous():
        raise ValueError("weight must be a contiguous [hidden] tensor of x's dtype")
    out = torch.empty((T, hidden), dtype=x.dtype, device=dev)
    total = None
    if delta is not None:
        if tuple(delta.shape) != (T, hidden) or delta.dtype != x.dtype or delta.stride(1) != 1:
            raise ValueError("delta must match x")
        total = torch.empty((T, hidden), dtype=x.dtype, device=dev)
    with _on(dev):
        rc = lib.mi355_rmsnorm(out.data_ptr(), total.data_ptr() if total is not None else None, x.data_ptr(),
                               _stride(x, 0), delta.data_ptr() if delta is not None else None,
                               _stride(delta, 0) if delta is not None else 0, weight.data_ptr(), float(eps),
                               T, hidden, dtype_code(x), _stream(dev))
    _lib.check(rc, 'mi355_rmsnorm')
    return out if delta is None else (total, out)


def silu_mul(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up over [T, width] (rows may be strided views of one projection output)."""
    dev = require_device(gate, up)
    lib = _lib.load()
    if gate.dim() != 2 or tuple(up.shape) != tuple(gate.shape) or gate.stride(1) != 1 or up.stride(1) != 1 \
            or up.dtype != gate.dtype:
        raise ValueError("gate and up must be [T, width] of one dtype with contiguous rows")
    T, width = gate.shape
    out = torch.empty((T, width), dtype=gate.dtype, device=dev)
    with _on(dev):
        rc = lib.mi355_silu_mul(out.data_ptr(), gate.data_ptr(), _stride(gate, 0), up.data_ptr(), _stride(up, 0),
                                T, width, dtype_code(gate), _stream(dev))
    _lib.check(rc, 'mi355_silu_mul')
    return out


def decode_step_inputs(packed: torch.Tensor, table_pool: torch.Tensor, block_size: int,
                       prev_ids: Optional[torch.Tensor] = None, gather_tables: bool = True):
    """packed int32 [3, rows] = (token, position, pool row) -> (token_ids i64 [rows], positions i64
    [rows], slots i64 [rows], context_lens i32 [rows], block_tables i32 [rows, width]).
    A token entry -1-j is replaced by prev_ids[j] (int32, the previous step's sampled ids).
    gather_tables=False skips the per-row table gather (block_tables is then None)."""
    dev = require_device(packed, table_pool, prev_ids)
    lib = _lib.load()
    if packed.dtype != torch.int32 or packed.dim() != 2 or packed.shape[0] != 3 or not packed.is_contiguous():
        raise ValueError("packed must be a contiguous int32 [3, rows] tensor")
    if table_pool.dtype != torch.int32 or table_pool.dim() != 2 or not table_pool.is_contiguous():
        raise ValueError("table_pool must be a contiguous int32 [pool_rows, width] tensor")
    if prev_ids is not None and (prev_ids.dtype != torch.int32 or prev_ids.dim() != 1 or not prev_ids.is_contiguous()):
        raise ValueError("prev_ids must be a contiguous int32 vector")
    rows, width = packed.shape[1], table_pool.shape[1]
    i64 = torch.empty((3, rows), dtype=torch.int64, device=dev)
    ctx = torch.empty((rows,), dtype=torch.int32, device=dev)
    tables = torch.empty((rows, width), dtype=torch.int32, device=dev) if gather_tables else None
    with _on(dev):
        rc = lib.mi355_decode_step_inputs(i64[0].data_ptr(), i64[1].data_ptr(), i64[2].data_ptr(), ctx.data_ptr(),
                                          tables.data_ptr() if gather_tables else None, packed.data_ptr(),
                                          table_pool.data_ptr(),
                                          prev_ids.data_ptr() if prev_ids is not None else None,
                                          prev_ids.numel() if prev_ids is not None else 0, rows,
                                          table_pool.shape[0], width, int(block_size), _stream(dev))
    _lib.check(rc, 'mi355_decode_step_inputs')
    return i64[0], i64[1], i64[2], ctx, tables


# MI355_PAGED_PREFILL=0 (read once, at import) sends chunks row by row through paged_decode (A/B measurements)
PAGED_PREFILL_ENABLED = __import__('os').environ.get('MI355_PAGED_PREFILL', '1') != '0'


def paged_prefill_supported(dtype: torch.dtype, head_dim: int) -> bool:
    """Shapes mi355_paged_prefill serves: 16-bit dtypes at head_dim 64 / 128 on the matrix-core
    kernels, fp32 and the other head dims on the VALU kernel (eight query rows per wave).  Anything
    else goes row by row through paged_decode."""
    return (PAGED_PREFILL_ENABLED and dtype in (torch.float32, torch.bfloat16, torch.float16)
            and head_dim in (16, 32, 64, 128, 256))


def paged_prefill_serves(dtype: torch.dtype, head_dim: int, block_size: int, n_q: int, context: int) -> bool:
    """Can a segment of n_q new tokens whose last one sits at position context - 1 be part of a
    paged_prefill launch?  The library answers (mi355_paged_prefill_max_context): its flash-attention
    kernels (chunks of >= 48 tokens; 16-bit head_dim 64 / 128, fp32 head_dim 32 / 64 / 128) read the
    block table from memory and have no context limit; the other kernels stage a sequence's whole
    table in LDS, which bounds the context at about 8190 blocks.  Segments beyond that go row by row
    through paged_decode, which splits long contexts."""
    if not paged_prefill_supported(dtype, head_dim):
        return False
    return context <= _lib.load().mi355_paged_prefill_max_context(_DTYPE_CODE[dtype], head_dim, block_size, n_q)


def paged_prefill(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_tables: torch.Tensor,
                  seg_q_start: torch.Tensor, seg_len: torch.Tensor, seg_pos0: torch.Tensor, max_q_len: int,
                  max_context: int, layout: str = 'NHSD', scale: Optional[float] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Causal attention of new query tokens over the paged cache (their K/V rows already written).

    q [T, Hq, D]; segment s owns rows seg_q_start[s] .. +seg_len[s]-1, its first token sits at
    position seg_pos0[s] and reads the cache through block_tables[s]; all three int32 [nseg].
    Rows of `out` [T, Hq, D] outside every segment are left untouched."""
    dev = require_device(q, k_cache, v_cache, block_tables, seg_q_start, seg_len, seg_pos0)
    lib = _lib.load()
    q = _head_rows(q)
    T, Hq, D = q.shape
    g = _kv_geometry(k_cache, layout)
    if tuple(v_cache.shape) != tuple(k_cache.shape) or _kv_geometry(v_cache, layout) != g:
        raise ValueError("k_cache and v_cache must share shape and strides")
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype or g['head_dim'] != D:
        raise TypeError("query and KV cache must share dtype and head_dim")
    for t in (block_tables, seg_q_start, seg_len, seg_pos0):
        if t.dtype != torch.int32:
            raise TypeError("block_tables and the segment vectors must be int32")
    nseg = seg_len.numel()
    block_tables = block_tables.contiguous()
    if block_tables.dim() != 2 or block_tables.shape[0] != nseg or seg_q_start.numel() != nseg \
            or seg_pos0.numel() != nseg:
        raise ValueError("block_tables must be [nseg, max_blocks]; the segment vectors [nseg]")
    out = _checked_out(out, (T, Hq, D), q.dtype, dev)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    with _on(dev):
        rc = lib.mi355_paged_prefill(
            out.data_ptr(), q.data_ptr(), _stride(q, 0), k_cache.data_ptr(), v_cache.data_ptr(),
            block_tables.data_ptr(), seg_q_start.contiguous().data_ptr(), seg_len.contiguous().data_ptr(),
            seg_pos0.contiguous().data_ptr(), nseg, int(max_q_len), Hq, g['heads'], D, g['block_size'],
            block_tables.shape[1], g['num_blocks'], int(max_context), g['s_block'], g['s_head'], g['s_tok'],
            float(scale), dtype_code(q), _stream(dev))
    _lib.check(rc, 'mi355_paged_prefill')
    return out
