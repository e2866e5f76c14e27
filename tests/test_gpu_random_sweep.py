"""Seeded random shape sweeps of the attention kernels against the numpy oracle.

The hand-picked cases in test_gpu_kernels.py aim at known borders (tile edges, split paths,
odd block sizes).  These sweeps draw every dimension at random -- batch, head grouping,
head_dim, block size, context lengths (zero included), dtype, cache layout -- so that
combinations nobody thought of still meet the oracle.  Each draw is small enough for the
oracle to finish in well under a second; seeds are fixed, so a failure names its case.
MI355_SWEEP_SCALE=n draws n times as many cases per sweep (a longer hunt, run by hand; the suite runs 1).
"""
import os

import numpy as np
import pytest
import torch

from oracle import paged_oracle as orc

pytestmark = pytest.mark.gpu

LOWP_TOL = dict(atol=1e-2, rtol=1.6e-2)
F32_TOL = dict(atol=1e-5, rtol=1e-5)
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def _seeds(n):
    return range(n * max(1, int(os.environ.get('MI355_SWEEP_SCALE', '1'))))


@pytest.fixture(scope='module')
def ops():
    from mi355_paged import ops as _ops
    return _ops


def _rounded(a, dtype):
    return a if dtype == torch.float32 else torch.from_numpy(a).to(dtype).float().numpy()


def _head_grouping(rng):
    hkv = int(rng.choice([1, 2, 3, 4, 8]))
    group = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 16]))
    return hkv * group, hkv


@pytest.mark.parametrize('seed', _seeds(48))
def test_decode_random_shapes(ops, device, seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = DTYPES[seed % 3]
    layout = 'NHSD' if (seed // 3) % 2 == 0 else 'NSHD'
    Hq, Hkv = _head_grouping(rng)
    D = int(rng.choice([16, 32, 64, 64, 128, 128, 256]))
    BS = int(rng.choice([1, 2, 3, 4, 8, 16, 16, 32, 64]))
    B = int(rng.integers(1, 9))
    max_ctx = int(rng.choice([1, 7, 33, 100, 260, 700, 1500, 4200] if B * Hq * D <= 4096 else [1, 7, 33, 100, 260, 700]))
    ctx = [int(c) for c in rng.integers(0, max_ctx + 1, size=B)]
    ctx[int(rng.integers(0, B))] = max_ctx
    per = (max_ctx + BS - 1) // BS
    NB = B * per + 3
    bt = rng.permutation(NB)[:B * per].reshape(B, per).astype(np.int32)
    cache = _rounded(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32), dtype)
    q = _rounded(rng.standard_normal((B, Hq, D), dtype=np.float32), dtype)

    tc = torch.from_numpy(cache).to(device=device, dtype=dtype)
    if layout == 'NSHD':
        tc = tc.permute(0, 1, 3, 2, 4).contiguous()
    out = ops.paged_decode(torch.from_numpy(q).to(device=device, dtype=dtype), tc[0], tc[1],
                           torch.from_numpy(bt).to(device), torch.tensor(ctx, dtype=torch.int32, device=device),
                           max_ctx, layout=layout)
    got = out.float().cpu().numpy()
    tol = F32_TOL if dtype == torch.float32 else LOWP_TOL
    for b in range(B):
        if ctx[b] == 0:
            assert (got[b] == 0).all(), (b, 'an empty context yields zeros')
            continue
        want = orc.paged_decode(q[b], cache[0], cache[1], bt[b].tolist(), ctx[b], BS, dtype=np.float64)
        np.testing.assert_allclose(got[b], want, err_msg=f'row {b} ctx {ctx[b]} {Hq}/{Hkv} D{D} BS{BS} {layout}', **tol)


@pytest.mark.parametrize('seed', _seeds(24))
def test_paged_prefill_random_segments(ops, device, seed):
    """Chunks of new tokens behind cached prefixes of random length; both the decode-shaped
    kernel (short chunks) and the flash-attention kernel in paged mode (48 tokens and up)."""
    rng = np.random.default_rng(2000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    layout = 'NHSD' if (seed // 2) % 2 == 0 else 'NSHD'
    Hq, Hkv = _head_grouping(rng)
    D = int(rng.choice([64, 128]))
    BS = int(rng.choice([1, 3, 8, 16, 16, 32, 128]))
    nseg = int(rng.integers(1, 5))
    segs = []
    for _ in range(nseg):
        n = int(rng.choice([1, 2, 15, 40, 47, 48, 49, 100, 130, 260]))
        p0 = int(rng.choice([0, 1, 15, 16, 31, 100, 127, 128, 500]))
        segs.append((p0, n))
    max_ctx = max(p0 + n for p0, n in segs)
    per = (max_ctx + BS - 1) // BS
    NB = nseg * per + 2
    bt = rng.permutation(NB)[:nseg * per].reshape(nseg, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(dtype)
    gap = int(rng.integers(0, 3))                                     # rows that belong to no segment
    T = sum(n for _, n in segs) + gap * (nseg + 1)
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    starts, q0 = [], gap
    for _, n in segs:
        starts.append(q0)
        q0 += n + gap
    tc = cache.to(device)
    if layout == 'NSHD':
        tc = tc.permute(0, 1, 3, 2, 4).contiguous()
    out = torch.full((T, Hq, D), 7.0, dtype=dtype, device=device)
    ops.paged_prefill(q.to(device), tc[0], tc[1], torch.from_numpy(bt).to(device),
                      torch.tensor(starts, dtype=torch.int32, device=device),
                      torch.tensor([n for _, n in segs], dtype=torch.int32, device=device),
                      torch.tensor([p0 for p0, _ in segs], dtype=torch.int32, device=device),
                      max(n for _, n in segs), max_ctx, layout=layout, out=out)
    got = out.float().cpu().numpy()
    qn, cn = q.float().numpy(), cache.float().numpy()
    owned = np.zeros(T, dtype=bool)
    for s, (p0, n) in enumerate(segs):
        rows = sorted(set([0, n - 1] + [int(j) for j in rng.integers(0, n, size=min(n, 6))]))
        owned[starts[s]:starts[s] + n] = True
        for j in rows:
            want = orc.paged_decode(qn[starts[s] + j], cn[0], cn[1], bt[s].tolist(), p0 + j + 1, BS, dtype=np.float64)
            np.testing.assert_allclose(got[starts[s] + j], want,
                                       err_msg=f'segment {s} {segs[s]} token {j} {Hq}/{Hkv} D{D} BS{BS} {layout}', **LOWP_TOL)
    assert (got[~owned] == 7.0).all()
    assert np.isfinite(got[owned]).all()


@pytest.mark.parametrize('seed', _seeds(24))
def test_prefill_causal_random_packing(ops, device, seed):
    """Packed variable-length prompts: fp32 runs the VALU kernel, 16-bit at head_dim 64 / 128 the
    matrix-core kernel, 16-bit at other head dims the VALU kernel again."""
    rng = np.random.default_rng(3000 + seed)
    dtype = DTYPES[seed % 3]
    Hq, Hkv = _head_grouping(rng)
    D = int(rng.choice([32, 64, 64, 128, 128, 256]))
    nseq = int(rng.integers(1, 6))
    lens = [int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 257, 400])) for _ in range(nseq)]
    if Hq * D > 2048:
        lens = [min(n, 130) for n in lens]
    T = sum(lens)
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    k = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    v = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = orc.varlen_causal_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, dtype=np.float64)
    out = ops.prefill_causal(q.to(device), k.to(device), v.to(device), torch.from_numpy(cu).to(device), max(lens))
    np.testing.assert_allclose(out.float().cpu().numpy(), want, err_msg=f'lens {lens} {Hq}/{Hkv} D{D}',
                               **(F32_TOL if dtype == torch.float32 else LOWP_TOL))


MLA_TOL = {torch.bfloat16: dict(atol=1e-2, rtol=1.6e-2), torch.float16: dict(atol=2e-3, rtol=4e-3)}


@pytest.mark.parametrize('seed', _seeds(24))
def test_mla_decode_random_shapes(ops, device, seed):
    """Latent-KV attention: kv_lora_rank, rope width, head count (both the 16-heads-per-workgroup
    and the 64-heads-per-workgroup kernels) and ragged contexts drawn at random."""
    from oracle import mla_oracle as mo
    rng = np.random.default_rng(4000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    R = int(rng.choice([32, 64, 128, 256, 512]))
    P = int(rng.choice([8, 16, 24, 32, 64]))
    H = int(rng.choice([1, 3, 8, 16, 17, 32, 40, 64, 65, 128]))
    seqs = int(rng.integers(1, 5))
    max_ctx = int(rng.choice([1, 15, 64, 65, 200, 513, 1100] if H * R <= 16384 else [1, 15, 64, 65, 200]))
    S = max_ctx + int(rng.integers(0, 4))                             # cache rows beyond the longest context
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(dtype)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[int(rng.integers(0, seqs))] = max_ctx
    cos, sin = mo.rope_tables(P, S + 1)
    scale = float(rng.choice([0.05, 0.11, (R // 4 + P) ** -0.5]))
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx, scale)
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), max_ctx, scale)
    np.testing.assert_allclose(out.cpu().numpy(), want, err_msg=f'H{H} R{R} P{P} ctx {ctx.tolist()}', **MLA_TOL[dtype])
    # the same rows scattered into blocks of a pool (block size and layout drawn too): the block-table
    # call returns the dense call's bits
    bs = int(rng.choice([1, 2, 8, 16, 64]))
    per = (S + bs - 1) // bs
    NB = seqs * per + 2
    bt = rng.permutation(NB)[:seqs * per].reshape(seqs, per).astype(np.int32)
    split = bool(rng.integers(0, 2))                     # a pool each for c_kv / k_rope, or one pool of R + P wide rows
    rows = torch.cat((c_kv, k_rope), dim=-1)
    rows = torch.nn.functional.pad(rows, (0, 0, 0, per * bs - S)).view(seqs * per, bs, R + P)
    pool = torch.full((NB, bs, R + P), float('nan'), dtype=dtype)
    pool[torch.from_numpy(bt.reshape(-1).astype(np.int64))] = rows
    pool = pool.to(device)
    pools = (pool[..., :R].contiguous(), pool[..., R:].contiguous()) if split else (pool[..., :R], pool[..., R:])
    paged = ops.mla_decode_paged(q.to(device), pools[0], pools[1], torch.from_numpy(bt).to(device),
                                 torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device),
                                 torch.from_numpy(ctx).to(device), max_ctx, scale)
    assert torch.equal(paged.view(torch.int32), out.view(torch.int32)), f'H{H} R{R} P{P} block {bs} split {split}'


@pytest.mark.parametrize('seed', _seeds(40))
def test_sampler_random_rows(ops, device, seed):
    """Greedy ids exact and the kept set / renormalised probabilities of the temperature, top-p and
    top-k filters against the sorting oracle, for random vocabulary sizes, row counts and settings."""
    rng = np.random.default_rng(5000 + seed)
    vocab = int(rng.choice([2, 17, 257, 1024, 4099, 32000, 32769, 50257, 100003]))
    B = int(rng.integers(1, 7))
    spread = float(rng.choice([0.5, 3.0, 8.0]))
    logits = (rng.standard_normal((B, vocab)) * spread).astype(np.float32)
    dev_logits = torch.from_numpy(logits).to(device)
    assert ops.sample_argmax(dev_logits).cpu().tolist() == logits.argmax(axis=1).tolist()
    temps = [float(t) for t in rng.choice([0.3, 0.7, 1.0, 1.5], size=B)]
    top_ps = [float(p) for p in rng.choice([0.2, 0.5, 0.9, 0.99, 1.0], size=B)]
    top_ks = [int(k) for k in rng.choice([0, 0, 1, 5, 50, 1000], size=B)]
    ids, probs = ops.sample_topk_topp(dev_logits, torch.tensor(temps, device=device), torch.tensor(top_ps, device=device),
                                      torch.tensor(top_ks, dtype=torch.int32, device=device), seed=seed, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(B):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        # only the token whose running mass lands within float rounding of top_p may differ; entries whose
        # probability underflows float32 (peaked rows at low temperature) are "kept" with probability 0
        diff = np.nonzero((got_kept != kept) & (np.maximum(want, probs[b]) > 1e-30))[0]
        if len(diff) > 1:
            # A flat row of a large vocabulary under a cut close to 1 keeps tens of thousands of entries of ~1e-6
            # each: the oracle's float32 running sum (the reference's, 05_mini_vllm/mini_vllm.py:621) is then only
            # good to ~1e-5 of mass, the kernel's sums are exact.  Entries may differ only where the mass above
            # them, summed in float64 over the float32 probabilities, lies that close to top_p.
            x = logits[b] / np.float32(temps[b])
            e = np.exp(x - x.max(), dtype=np.float32)
            p64 = (e / e.sum(dtype=np.float32)).astype(np.float64)
            order = np.argsort(-p64, kind='stable')
            above = np.empty(vocab)
            above[order] = np.cumsum(p64[order]) - p64[order]
            assert top_ks[b] == 0 and np.all(np.abs(above[diff] - top_ps[b]) < 3e-5), \
                f'row {b} vocab {vocab} T{temps[b]} p{top_ps[b]} k{top_ks[b]}: kept sets differ at {diff}'
            continue
        if len(diff) == 0:
            np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-7)
        assert got_kept[int(ids[b])]
        assert abs(probs[b].sum() - 1.0) < 1e-4


def _serve(mv, seed, graphs, lookahead):
    """One randomly drawn serving run: engine limits, prompts (some sharing a prefix), generation
    lengths and the step at which each request arrives all come from the seed."""
    rng = np.random.default_rng(6000 + seed)
    vocab = 12                                           # id 0 ends a sequence: sampled often, at unpredictable steps
    block_size = int(rng.choice([4, 8, 16]))
    slots = int(rng.choice([3, 8, 16]))
    budget = int(rng.choice([24, 64, 256]))
    prefix_caching = bool(rng.integers(0, 2))
    n = int(rng.integers(6, 36))
    shared = [int(t) for t in rng.integers(1, vocab, size=int(rng.integers(block_size, 3 * block_size + 1)))]
    prompts = []
    for _ in range(n):
        own = [int(t) for t in rng.integers(1, vocab, size=int(rng.integers(1, 50)))]
        prompts.append(shared + own if rng.random() < 0.4 else own)
    max_new = [int(m) for m in rng.integers(1, 36, size=n)]
    arrival = sorted(int(a) for a in np.where(rng.random(n) < 0.5, 0, rng.integers(0, 60, size=n)))
    torch.manual_seed(7)
    eng = mv.MiniVLLM(vocab_size=vocab, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                      num_kv_blocks=1024, block_size=block_size, max_num_seqs=slots, max_num_batched_tokens=budget,
                      device='cuda', dtype=torch.float32, decode_graphs=graphs, decode_lookahead=lookahead,
                      enable_prefix_caching=prefix_caching)
    outs, ids, step, nxt = {}, [], 0, 0
    while nxt < n or eng.scheduler.has_unfinished:
        while nxt < n and arrival[nxt] <= step:
            ids.append(eng.add_request(prompts[nxt], max_new_tokens=max_new[nxt], temperature=0.0))
            nxt += 1
        for rid, tok in eng.step():
            outs.setdefault(rid, []).append(tok)
        step += 1
        assert step < 5000, 'the run does not end'
    return eng, [outs.get(rid, []) for rid in ids], max_new


@pytest.mark.parametrize('seed', _seeds(10))
def test_engine_random_serving_runs(device, seed):
    """Requests that arrive while others run, more requests than sequence slots, token budgets that
    chunk the prompts, shared prefixes with and without prefix caching: the engine that replays
    captured decode steps, and the one that also issues them ahead, give every request exactly the
    tokens of the engine that runs every step eagerly (greedy, fp32)."""
    import mini_vllm as mv
    eager, out_e, max_new = _serve(mv, seed, False, False)
    plain, out_p, _ = _serve(mv, seed, True, False)
    ahead, out_a, _ = _serve(mv, seed, True, True)
    for toks, m in zip(out_e, max_new):
        assert 1 <= len(toks) <= m and (len(toks) == m or toks[-1] == 0)
    assert out_p == out_e
    assert out_a == out_e
    assert ahead._inflight is None
    for eng in (eager, plain, ahead):
        assert eng.allocator.num_free == 1024
        assert not eng.scheduler.running and not eng.scheduler.waiting


LAYER_TOL = {torch.float32: dict(atol=1e-5, rtol=1e-5), torch.bfloat16: dict(atol=2e-2, rtol=2e-2),
             torch.float16: dict(atol=3e-3, rtol=3e-3)}


def _rotate_half(x):
    lo, hi = x.chunk(2, dim=-1)
    return torch.cat((-hi, lo), dim=-1)


@pytest.mark.parametrize('seed', _seeds(18))
def test_layer_steps_random_shapes(ops, device, seed):
    """RMSNorm (+ carried residual), the SwiGLU gate and RoPE + KV write on randomly drawn widths, row
    counts, head geometries, block sizes and cache layouts, against the torch expressions."""
    g = torch.Generator().manual_seed(7000 + seed)
    rnd = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    dtype = DTYPES[seed % 3]
    per16 = 16 // torch.empty((), dtype=dtype).element_size()
    tol = LAYER_TOL[dtype]

    # rmsnorm over [T, hidden] rows that are a column slice of a wider activation
    T, hidden = rnd(1, 300), per16 * rnd(1, 4096 // per16 // (2 if dtype == torch.float32 else 1))
    wide = (torch.randn(T, hidden + per16 * rnd(0, 3), generator=g) * 3).to(dtype).to(device)
    x = wide[:, :hidden]
    delta = torch.randn(T, hidden, generator=g).to(dtype).to(device)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).to(device)
    eps = [1e-6, 1e-5][seed % 2]

    def norm(t):
        t = t.float()
        return w.float() * (t / (t * t).mean(-1, keepdim=True).add(eps).sqrt())

    torch.testing.assert_close(ops.rmsnorm(x, w, eps).float(), norm(x), **tol)
    total, normed = ops.rmsnorm(x, w, eps, delta=delta)
    assert torch.equal(total, (x.float() + delta.float()).to(dtype))
    torch.testing.assert_close(normed.float(), norm(total), **tol)

    # silu(gate) * up on the two halves of one projection output
    width = per16 * rnd(1, 400)
    both = (torch.randn(T, 2 * width, generator=g) * 4).to(dtype).to(device)
    torch.testing.assert_close(ops.silu_mul(both[:, :width], both[:, width:]).float(),
                               torch.nn.functional.silu(both[:, :width].float()) * both[:, width:].float(), **tol)

    # RoPE in place on q and k (views of one qkv projection) + write of k, v into the paged caches
    Hkv = [1, 2, 4, 8][rnd(0, 3)]
    Hq = Hkv * [1, 2, 4][rnd(0, 2)]
    D = [16, 32, 64, 128][rnd(0, 3)]
    BS = [1, 3, 4, 16, 32][rnd(0, 4)]
    layout = ['NSHD', 'NHSD'][seed % 2]
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, generator=g).to(dtype).to(device)
    q, k, v = (t.view(T, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.outer(torch.arange(700).float(), inv)
    cos, sin = (torch.cat([f(ang), f(ang)], -1).to(dtype).to(device) for f in (torch.cos, torch.sin))
    pos = torch.randint(0, 700, (T,), generator=g).to(device)
    NB = (T + BS - 1) // BS + 2
    slots = torch.randperm(NB * BS, generator=g)[:T].to(device)
    slots[rnd(0, T - 1)] = -1                                  # a row that is rotated but not stored
    shape = (NB, BS, Hkv, D) if layout == 'NSHD' else (NB, Hkv, BS, D)
    kc = torch.zeros(shape, dtype=dtype, device=device)
    vc = torch.zeros_like(kc)
    c, s = cos[pos].float().unsqueeze(1), sin[pos].float().unsqueeze(1)
    want_q, want_k = q.float() * c + _rotate_half(q.float()) * s, k.float() * c + _rotate_half(k.float()) * s
    v0 = v.clone()
    ops.rope_kv_write(q, k, v, cos, sin, pos, slots, kc, vc, layout=layout)
    torch.testing.assert_close(q.float(), want_q, **tol)
    torch.testing.assert_close(k.float(), want_k, **tol)
    assert torch.equal(v, v0)
    flat = (lambda t: t.reshape(NB * BS, Hkv, D)) if layout == 'NSHD' else \
        (lambda t: t.permute(0, 2, 1, 3).reshape(NB * BS, Hkv, D))
    keep = slots >= 0
    assert torch.equal(flat(kc)[slots[keep]], k[keep]) and torch.equal(flat(vc)[slots[keep]], v[keep])
    untouched = torch.ones(NB * BS, dtype=torch.bool, device=device)
    untouched[slots[keep]] = False
    assert not flat(kc)[untouched].any() and not flat(vc)[untouched].any()

    # the plain slot write gives the same cache contents
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.kv_write_slots(kc2, vc2, k[keep], v[keep], slots[keep], layout=layout)
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc)
