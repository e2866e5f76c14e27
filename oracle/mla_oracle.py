"""CPU oracle for Multi-head Latent Attention.

TEST INFRASTRUCTURE ONLY (never imported by the product).  Parity status: PINNED --
checked against outputs captured from the reference module (tests/golden/g9_mla.npz,
tests/test_oracle_golden.py).

`mla_forward` restates MultiHeadLatentAttention.forward (04_mla/mini_mla.py:203-306) in
numpy, step for step (expanded K/V, rope applied to cached k_rope at positions 0..S-1,
additive causal mask with diagonal S-T+1).  `latent_attention` is the algebraically equal
weight-absorbed form the HIP kernel computes; `absorb_query` / `expand_output` are the two
dense products that surround it.
"""
from __future__ import annotations

import numpy as np


def rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def rope_tables(dim, length, base=10000.0, dtype=np.float32):
    """cos/sin [length, dim] as 04_mla/mini_mla.py:50-66 builds them (float32 arithmetic)."""
    inv_freq = (1.0 / (np.float32(base) ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
    freqs = np.outer(np.arange(length, dtype=np.float32), inv_freq).astype(np.float32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(dtype), np.sin(emb).astype(dtype)


def rms_norm(x, weight, eps=None):
    """torch.nn.RMSNorm with eps=None: eps = machine epsilon of the input dtype."""
    if eps is None:
        eps = np.finfo(x.dtype).eps
    return x / np.sqrt((x * x).mean(axis=-1, keepdims=True) + eps) * weight


def _softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def mla_forward(w, cfg, hidden, positions, past_c_kv=None, past_k_rope=None, dtype=np.float32):
    """w: dict of weight arrays keyed like the module's state_dict; cfg: dict with num_heads,
    qk_nope_head_dim, qk_rope_head_dim, kv_lora_rank, v_head_dim.  Returns (out, new_c_kv, new_k_rope)."""
    H, nope, rope, lora, vd = (cfg[k] for k in ('num_heads', 'qk_nope_head_dim', 'qk_rope_head_dim',
                                                'kv_lora_rank', 'v_head_dim'))
    x = hidden.astype(dtype)
    B, T, _ = x.shape
    W = {k: v.astype(dtype) for k, v in w.items()}
    q = rms_norm(x @ W['q_a_proj.weight'].T, W['q_a_layernorm.weight']) @ W['q_b_proj.weight'].T
    q = q.reshape(B, T, H, nope + rope)
    q_nope, q_rope = q[..., :nope], q[..., nope:]
    kvc = x @ W['kv_a_proj_with_mqa.weight'].T
    new_c_kv = rms_norm(kvc[..., :lora], W['kv_a_layernorm.weight'])
    new_k_rope = kvc[..., lora:]
    if past_c_kv is not None:
        c_kv = np.concatenate([past_c_kv.astype(dtype), new_c_kv], axis=1)
        k_rope = np.concatenate([past_k_rope.astype(dtype), new_k_rope], axis=1)
    else:
        c_kv, k_rope = new_c_kv, new_k_rope
    S = c_kv.shape[1]
    kv = (c_kv @ W['kv_b_proj.weight'].T).reshape(B, S, H, nope + vd)
    k_nope, v = kv[..., :nope], kv[..., nope:]
    cos, sin = rope_tables(rope, max(int(positions.max()) + 1, S), dtype=dtype)
    cq, sq = cos[positions][:, :, None, :], sin[positions][:, :, None, :]
    q_rope = q_rope * cq + rotate_half(q_rope) * sq
    ck, sk = cos[:S][None, :, None, :], sin[:S][None, :, None, :]
    k_r = np.broadcast_to(k_rope[:, :, None, :], (B, S, H, rope))
    k_r = k_r * ck + rotate_half(k_r) * sk
    q_full = np.concatenate([q_nope, q_rope], axis=-1).transpose(0, 2, 1, 3)
    k_full = np.concatenate([k_nope, k_r], axis=-1).transpose(0, 2, 1, 3)
    scale = dtype((nope + rope) ** -0.5)
    attn = q_full @ k_full.transpose(0, 1, 3, 2) * scale
    if T > 1:
        mask = np.triu(np.full((T, S), -np.inf, dtype=dtype), k=S - T + 1)
        attn = attn + mask
    out = _softmax(attn) @ v.transpose(0, 2, 1, 3)
    out = out.transpose(0, 2, 1, 3).reshape(B, T, H * vd)
    return out @ W['o_proj.weight'].T, new_c_kv, new_k_rope


def absorb_query(q_nope, w_kv_b, H, nope, vd):
    """q_lat[..., h, :] = W_k[h]^T q_nope[..., h, :], W_k[h] = rows of kv_b_proj for head h's k_nope."""
    lora = w_kv_b.shape[1]
    wk = w_kv_b.reshape(H, nope + vd, lora)[:, :nope, :]
    return np.einsum('...hn,hnl->...hl', q_nope, wk)


def expand_output(o_lat, w_kv_b, H, nope, vd):
    lora = w_kv_b.shape[1]
    wv = w_kv_b.reshape(H, nope + vd, lora)[:, nope:, :]
    return np.einsum('...hl,hvl->...hv', o_lat, wv)


def latent_attention(q_cat, c_kv, k_rope, cos, sin, context_lens, scale, row_seq=None, dtype=np.float64):
    """The absorbed core: q_cat [rows, H, R+P]; c_kv [seqs, S, R]; k_rope [seqs, S, P] un-rotated;
    cos/sin [len, P].  Row r attends to tokens 0..context_lens[r]-1 of sequence row_seq[r]."""
    rows, H, _ = q_cat.shape
    R = c_kv.shape[2]
    out = np.zeros((rows, H, R), dtype=dtype)
    for r in range(rows):
        n = int(context_lens[r])
        if n <= 0:
            continue
        s = r if row_seq is None else int(row_seq[r])
        c = c_kv[s, :n].astype(dtype)
        kr = k_rope[s, :n].astype(dtype)
        kr = kr * cos[:n].astype(dtype) + rotate_half(kr) * sin[:n].astype(dtype)
        q = q_cat[r].astype(dtype)
        sc = (q[:, :R] @ c.T + q[:, R:] @ kr.T) * dtype(scale)
        out[r] = _softmax(sc) @ c
    return out


def latent_attention_paged(q_cat, pool_c_kv, pool_k_rope, block_tables, block_size, cos, sin, context_lens, scale,
                           row_seq=None, dtype=np.float64):
    """latent_attention over a block cache (gather form): pool_c_kv [blocks, block_size, R] and
    pool_k_rope [blocks, block_size, P]; token t of sequence s is row t % block_size of block
    block_tables[s][t // block_size].  The rows of every sequence are gathered into dense arrays
    (what MLAKVCache would hold, 04_mla/mini_mla.py:79-132) and handed to latent_attention."""
    block_tables = np.asarray(block_tables)
    seqs = block_tables.shape[0]
    longest = int(max(context_lens)) if len(context_lens) else 0
    R, P = pool_c_kv.shape[2], pool_k_rope.shape[2]
    c_kv = np.zeros((seqs, max(longest, 1), R), dtype=pool_c_kv.dtype)
    k_rope = np.zeros((seqs, max(longest, 1), P), dtype=pool_k_rope.dtype)
    need = np.zeros(seqs, dtype=np.int64)
    for r, n in enumerate(context_lens):
        s = r if row_seq is None else int(row_seq[r])
        need[s] = max(need[s], int(n))
    for s in range(seqs):
        for t in range(int(need[s])):
            blk = int(block_tables[s][t // block_size])
            c_kv[s, t] = pool_c_kv[blk, t % block_size]
            k_rope[s, t] = pool_k_rope[blk, t % block_size]
    return latent_attention(q_cat, c_kv, k_rope, cos, sin, context_lens, scale, row_seq, dtype)
