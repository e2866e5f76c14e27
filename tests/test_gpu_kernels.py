"""HIP kernels vs the CPU oracle and the golden vectors (needs an MI355X).

Every call goes through the C ABI (ctypes binding in mi355_paged/_lib.py).
Tolerances:
  fp32 kernels:  atol=1e-5, rtol=1e-5 against the fp32 golden outputs (the reference's own bar,
                 01_paged_attention/test_paged_attention.py:99);
  bf16 / fp16:   oracle = float64 attention on the rounded inputs; the kernel accumulates in
                 fp32 and rounds once on store -> atol=1e-2, rtol=1.6e-2 (two bf16 ulps);
  indices, slot writes, greedy ids: bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_json, load_npz, npz_cases
from oracle import paged_oracle as orc

pytestmark = pytest.mark.gpu

LOWP_TOL = dict(atol=1e-2, rtol=1.6e-2)
F32_TOL = dict(atol=1e-5, rtol=1e-5)


@pytest.fixture(scope='module')
def ops():
    from mi355_paged import ops as _ops
    return _ops


def _decode_on_gpu(ops, device, q, cache, block_tables, ctx, dtype, layout='NHSD'):
    """q [B,Hq,D], cache [2,NB,Hkv,BS,D] float32 numpy -> numpy float32 result."""
    tq = torch.from_numpy(q).to(device=device, dtype=dtype)
    tc = torch.from_numpy(cache).to(device=device, dtype=dtype)
    if layout == 'NSHD':
        tc = tc.permute(0, 1, 3, 2, 4).contiguous()
    bt = torch.from_numpy(np.asarray(block_tables, dtype=np.int32)).to(device)
    cl = torch.tensor(ctx, dtype=torch.int32, device=device)
    out = ops.paged_decode(tq, tc[0], tc[1], bt, cl, max(ctx) if len(ctx) else 0, layout=layout)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def test_g1_known_answer_fp32(ops, device):
    z = load_npz('g1_decode_known_answer.npz')
    out = _decode_on_gpu(ops, device, z['query'][None], z['kv_cache'], [z['block_table'].tolist()],
                         [int(z['seq_len'])], torch.float32)[0]
    np.testing.assert_allclose(out, z['out_paged'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out, z['out_dense'], atol=1e-5, rtol=0)


@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
@pytest.mark.parametrize('case', npz_cases('g2_decode_cases.npz'), ids=lambda c: str(c[0]))
def test_g2_decode_cases_fp32(ops, device, case, layout):
    m, f = case
    out = _decode_on_gpu(ops, device, f['query'].astype(np.float32)[None], f['kv_cache'].astype(np.float32),
                         [f['block_table'].tolist()], [m['s']], torch.float32, layout)[0]
    np.testing.assert_allclose(out, f['out'], **F32_TOL)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', npz_cases('g2_decode_cases.npz'), ids=lambda c: str(c[0]))
def test_g2_decode_cases_lowp(ops, device, case, dtype):
    m, f = case
    q = torch.from_numpy(f['query'].astype(np.float32)).to(dtype).float().numpy()
    cache = torch.from_numpy(f['kv_cache'].astype(np.float32)).to(dtype).float().numpy()
    want = orc.paged_decode(q, cache[0], cache[1], f['block_table'].tolist(), m['s'], m['bs'], dtype=np.float64)
    out = _decode_on_gpu(ops, device, q[None], cache, [f['block_table'].tolist()], [m['s']], dtype)[0]
    np.testing.assert_allclose(out, want, **LOWP_TOL)


def _random_batch(rng, B, Hq, Hkv, D, BS, max_ctx, ragged=True, spare=7):
    ctx = rng.integers(1, max_ctx + 1, size=B) if ragged else np.full(B, max_ctx)
    per = (max_ctx + BS - 1) // BS
    NB = B * per + spare
    perm = rng.permutation(NB)[:B * per].reshape(B, per).astype(np.int32)
    cache = rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)
    q = rng.standard_normal((B, Hq, D), dtype=np.float32)
    return q, cache, perm, [int(c) for c in ctx]


@pytest.mark.parametrize('shape', [
    # B, Hq, Hkv, D, BS, max_ctx
    (5, 8, 8, 64, 16, 300), (3, 8, 4, 64, 16, 700), (4, 8, 2, 32, 8, 130), (2, 8, 1, 128, 16, 257),
    (3, 12, 4, 64, 16, 90), (2, 16, 1, 64, 16, 70), (3, 4, 4, 16, 4, 50), (2, 6, 6, 256, 16, 40),
    (7, 8, 8, 64, 3, 100), (1, 8, 8, 64, 16, 5000),
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_decode_batched_ragged(ops, device, shape, dtype):
    """Ragged batches, GQA groupings, odd block sizes and long contexts (split-KV path)."""
    B, Hq, Hkv, D, BS, max_ctx = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    q, cache, bt, ctx = _random_batch(rng, B, Hq, Hkv, D, BS, max_ctx)
    if dtype != torch.float32:
        q = torch.from_numpy(q).to(dtype).float().numpy()
        cache = torch.from_numpy(cache).to(dtype).float().numpy()
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, BS, dtype=np.float64)
    out = _decode_on_gpu(ops, device, q, cache, bt, ctx, dtype)
    np.testing.assert_allclose(out, want, **(F32_TOL if dtype == torch.float32 else LOWP_TOL))


@pytest.mark.parametrize('shape', [
    # B, Hq, Hkv, D, BS, max_ctx   -- block sizes from 1 to 128, many kv heads, wide query groups
    (3, 8, 8, 64, 1, 40), (3, 8, 8, 64, 2, 77), (2, 8, 8, 64, 32, 200), (2, 4, 4, 128, 64, 300),
    (2, 8, 8, 32, 128, 513), (2, 16, 16, 64, 16, 150), (2, 32, 32, 32, 16, 90), (2, 32, 2, 64, 16, 120),
    (2, 24, 24, 16, 16, 65), (300, 8, 8, 64, 16, 48),
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_decode_shape_sweep(ops, device, shape, dtype):
    B, Hq, Hkv, D, BS, max_ctx = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    q, cache, bt, ctx = _random_batch(rng, B, Hq, Hkv, D, BS, max_ctx)
    ctx[0] = max_ctx                                   # one row ends exactly on max_context
    if B > 1:
        ctx[1] = max(1, (max_ctx // BS) * BS)          # one ends exactly on a block boundary
    if dtype != torch.float32:
        q = torch.from_numpy(q).to(dtype).float().numpy()
        cache = torch.from_numpy(cache).to(dtype).float().numpy()
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, BS, dtype=np.float64)
    out = _decode_on_gpu(ops, device, q, cache, bt, ctx, dtype)
    np.testing.assert_allclose(out, want, **(F32_TOL if dtype == torch.float32 else LOWP_TOL))


@pytest.mark.parametrize('shape', [
    # B, Hq, Hkv, D, BS, max_ctx -- grouped-query shapes served by the matrix-core decode kernel
    (3, 32, 8, 128, 16, 300),      # 4 query heads per kv head, 4 kv heads per workgroup
    (2, 64, 8, 128, 16, 1100),     # 8 per kv head
    (4, 16, 2, 64, 16, 700),       # 2 kv heads: waves take token sub-ranges (LDS merge)
    (5, 8, 1, 64, 16, 900),        # 1 kv head, 4 sub-ranges
    (2, 16, 1, 64, 8, 5000),       # long context: token splits + merge kernel
    (3, 20, 4, 64, 16, 200),       # 5 query heads per kv head (partial head tile)
    (2, 40, 2, 128, 32, 333),      # 20 per kv head: two head tiles
    (3, 24, 6, 64, 3, 150),        # block size that is not a power of two
    (2, 16, 4, 64, 1, 70),         # one token per block
    (130, 32, 4, 64, 16, 96),      # many short sequences
])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
def test_decode_grouped_query_matrix_core(ops, device, shape, dtype, layout):
    B, Hq, Hkv, D, BS, max_ctx = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    q, cache, bt, ctx = _random_batch(rng, B, Hq, Hkv, D, BS, max_ctx)
    ctx[0] = max_ctx
    ctx[-1] = 1                                        # a single-token context
    if B > 2:
        ctx[1] = max(1, (max_ctx // 32) * 32)          # ends exactly on a pipeline-step boundary
    q = torch.from_numpy(q).to(dtype).float().numpy()
    cache = torch.from_numpy(cache).to(dtype).float().numpy()
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, BS, dtype=np.float64)
    out = _decode_on_gpu(ops, device, q, cache, bt, ctx, dtype, layout)
    np.testing.assert_allclose(out, want, **LOWP_TOL)


def test_decode_grouped_query_matches_valu_kernel(ops, device):
    """Same inputs through the matrix-core kernel and (knob MI355_DECODE_GQA_MFMA = 0) the VALU kernel:
    both within tolerance of the oracle and of each other; zero-length rows give zeros in both."""
    rng = np.random.default_rng(77)
    B, Hq, Hkv, D, BS, max_ctx = 6, 32, 4, 128, 16, 400
    q, cache, bt, ctx = _random_batch(rng, B, Hq, Hkv, D, BS, max_ctx)
    ctx[2] = 0
    q = torch.from_numpy(q).to(torch.bfloat16).float().numpy()
    cache = torch.from_numpy(cache).to(torch.bfloat16).float().numpy()
    mfma = _decode_on_gpu(ops, device, q, cache, bt, ctx, torch.bfloat16)
    with ops.debug_knob('MI355_DECODE_GQA_MFMA', 0):
        valu = _decode_on_gpu(ops, device, q, cache, bt, ctx, torch.bfloat16)
    np.testing.assert_allclose(mfma, valu, **LOWP_TOL)
    assert not mfma[2].any() and not valu[2].any()
    keep = [i for i in range(B) if ctx[i] > 0]
    want = orc.paged_decode_batch(q[keep], cache[0], cache[1], bt[keep], [ctx[i] for i in keep], BS, dtype=np.float64)
    np.testing.assert_allclose(mfma[keep], want, **LOWP_TOL)


@pytest.mark.parametrize('shape', [
    # Hq, Hkv, D, BS, [(pos0, n_query_tokens), ...]
    (8, 8, 64, 16, [(0, 40), (130, 17), (5, 1), (64, 64)]),          # one head per kv head: 16 tokens per tile
    (8, 4, 64, 16, [(33, 50), (0, 7)]),                              # 2 heads per kv head: 8 tokens x 2 heads
    (16, 4, 128, 16, [(100, 35), (7, 3), (256, 16)]),                # 4 x 4
    (16, 2, 64, 8, [(1000, 21)]),                                    # 8 heads per kv head, long prefix, sub-ranges
    (12, 4, 64, 16, [(20, 30)]),                                     # 3 heads per kv head -> padded to 4
    (40, 2, 128, 32, [(50, 9), (0, 33)]),                            # 20 heads per kv head: two head tiles, 1 token
    (4, 4, 64, 3, [(10, 29)]),                                       # block size not a power of two
    # chunks of 48 or more query tokens go to the flash-attention kernel (128 query rows per workgroup)
    (8, 8, 64, 16, [(0, 300), (130, 257), (1000, 129), (77, 1)]),    # no prefix; prefix not on a tile border
    (16, 4, 128, 16, [(200, 140), (0, 64)]),                         # grouped heads, head_dim 128
    (4, 2, 64, 3, [(10, 70)]),                                       # block size not a power of two
    (8, 8, 128, 32, [(31, 200)]),                                    # block size 32
    (4, 4, 64, 128, [(100, 130)]),                                   # a block holds two key tiles
])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
def test_paged_prefill_segments(ops, device, shape, dtype, layout):
    """Several new query tokens per sequence over the paged cache: token j of a segment attends to
    positions 0 .. pos0 + j.  Checked token by token against the decode oracle with that context."""
    Hq, Hkv, D, BS, segs = shape
    rng = np.random.default_rng(hash((Hq, Hkv, D, BS, len(segs))) % (2 ** 31))
    nseg = len(segs)
    max_ctx = max(p0 + n for p0, n in segs)
    per = (max_ctx + BS - 1) // BS
    NB = nseg * per + 5
    bt = rng.permutation(NB)[:nseg * per].reshape(nseg, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(dtype)
    T = sum(n for _, n in segs) + 3                                   # 3 rows that belong to no segment
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    q0, starts = 2, []                                                # segments start after 2 foreign rows
    for _, n in segs:
        starts.append(q0)
        q0 += n
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
        for j in range(n):
            row = starts[s] + j
            owned[row] = True
            want = orc.paged_decode(qn[row], cn[0], cn[1], bt[s].tolist(), p0 + j + 1, BS, dtype=np.float64)
            np.testing.assert_allclose(got[row], want, **LOWP_TOL)
    assert (got[~owned] == 7.0).all()                                 # rows outside every segment are untouched


@pytest.fixture
def prefill_waves(ops, request):
    with ops.debug_knob('MI355_PREFILL_WAVES', int(request.param)):
        yield request.param


@pytest.mark.parametrize('prefill_waves', ['4', '8'], indirect=True)
def test_paged_prefill_long_chunk_matches_packed_prefill(ops, device, prefill_waves):
    """A whole prompt presented as one chunk over the paged cache (no prefix) must equal causal
    prompt attention over the same K/V rows; with a prefix, the oracle row by row on a sample.
    Both workgroup widths of the kernel."""
    rng = np.random.default_rng(int(prefill_waves))
    Hq, Hkv, D, BS, n = 8, 4, 128, 16, 700
    per = (n + 333 + BS - 1) // BS
    NB = 2 * per + 3
    bt = rng.permutation(NB)[:2 * per].reshape(2, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(torch.bfloat16)
    q = torch.from_numpy(rng.standard_normal((2 * n, Hq, D), dtype=np.float32)).to(torch.bfloat16)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    tc = cache.to(device)
    out = ops.paged_prefill(q.to(device), tc[0], tc[1], torch.from_numpy(bt).to(device), i32([0, n]), i32([n, n]),
                            i32([0, 333]), n, n + 333)
    got = out.float().cpu().numpy()
    # segment 0: gather its K/V rows and run the packed causal kernel's oracle
    cn = cache.float().numpy()
    rows = [bt[0][t // BS] for t in range(n)]
    k = np.stack([cn[0][rows[t], :, t % BS, :] for t in range(n)])
    v = np.stack([cn[1][rows[t], :, t % BS, :] for t in range(n)])
    want = orc.varlen_causal_attention(q[:n].float().numpy(), k, v, np.array([0, n], dtype=np.int32), dtype=np.float64)
    np.testing.assert_allclose(got[:n], want, **LOWP_TOL)
    qn = q.float().numpy()
    for j in [0, 1, 63, 64, 127, 128, 255, 256, 511, 699]:
        want = orc.paged_decode(qn[n + j], cn[0], cn[1], bt[1].tolist(), 333 + j + 1, BS, dtype=np.float64)
        np.testing.assert_allclose(got[n + j], want, **LOWP_TOL)


def test_paged_prefill_clamps_bad_block_ids(ops, device):
    """Block ids outside the pool inside the used range give a wrong answer for that segment but
    no out-of-bounds access and no effect on the other segment."""
    rng = np.random.default_rng(5)
    Hq, D, BS, n = 4, 64, 16, 100
    per = (n + BS - 1) // BS
    NB = 2 * per
    bt = rng.permutation(NB).reshape(2, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hq, BS, D), dtype=np.float32)).to(torch.bfloat16).to(device)
    q = torch.from_numpy(rng.standard_normal((2 * n, Hq, D), dtype=np.float32)).to(torch.bfloat16).to(device)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    good = ops.paged_prefill(q, cache[0], cache[1], torch.from_numpy(bt).to(device), i32([0, n]), i32([n, n]),
                             i32([0, 0]), n, n).clone()
    bad = bt.copy()
    bad[0, 2], bad[0, 4] = 10 ** 8, -5
    got = ops.paged_prefill(q, cache[0], cache[1], torch.from_numpy(bad).to(device), i32([0, n]), i32([n, n]),
                            i32([0, 0]), n, n)
    torch.cuda.synchronize()
    assert torch.equal(got[n:], good[n:])
    assert torch.isfinite(got.float()).all()


@pytest.mark.parametrize('n', [100, 20])         # flash-attention kernel (>= 48 query tokens) and the append kernel
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_paged_prefill_survives_poisoned_segments(ops, device, n, dtype):
    """Segment descriptors that contradict the host-side bounds (a start position far past the
    table, a negative one, a length beyond max_q_len) must neither fault nor disturb the healthy
    segment: no key at or past max_context is read in any of the three back ends."""
    rng = np.random.default_rng(n)
    Hq, D, BS = 4, 64, 16
    per = (n + BS - 1) // BS
    NB = 3 * per
    bt = rng.permutation(NB).reshape(3, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hq, BS, D), dtype=np.float32)).to(dtype).to(device)
    q = torch.from_numpy(rng.standard_normal((3 * n, Hq, D), dtype=np.float32)).to(dtype).to(device)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    tables = torch.from_numpy(bt).to(device)
    good = ops.paged_prefill(q, cache[0], cache[1], tables, i32([0, n, 2 * n]), i32([n, n, n]), i32([0, 0, 0]),
                             n, n).clone()
    got = ops.paged_prefill(q, cache[0], cache[1], tables, i32([0, n, 2 * n]), i32([n, n, n]),
                            i32([10 ** 9, 0, -7]), n, n)
    torch.cuda.synchronize()
    assert torch.equal(got[n:2 * n], good[n:2 * n])
    got = ops.paged_prefill(q, cache[0], cache[1], tables, i32([0, n, 2 * n]), i32([n, n, 10 ** 6]),
                            i32([n // 2, 0, 0]), n, n)
    torch.cuda.synchronize()
    assert torch.equal(got[n:2 * n], good[n:2 * n])
    assert torch.equal(got[2 * n:], good[2 * n:])     # rows past max_q_len do not exist: the segment's n rows are as before


def test_paged_prefill_rejects_unsupported_shapes(ops, device):
    from mi355_paged._lib import Mi355Error
    assert ops.paged_prefill_supported(torch.bfloat16, 64) and ops.paged_prefill_supported(torch.float32, 32)
    assert not ops.paged_prefill_supported(torch.float32, 48)
    q = torch.zeros(4, 2, 48, dtype=torch.bfloat16, device=device)
    cache = torch.zeros(2, 2, 16, 48, dtype=torch.bfloat16, device=device)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    with pytest.raises(Mi355Error, match='head_dim'):
        ops.paged_prefill(q, cache, cache, i32([[0, 1]]), i32([0]), i32([4]), i32([0]), 4, 4)


@pytest.mark.parametrize('case', [(torch.float32, 16, 8), (torch.bfloat16, 64, 8)])
def test_paged_prefill_serves_the_context_its_query_function_names(ops, device, case):
    """mi355_paged_prefill_max_context is the longest context a launch of short segments takes: a
    segment that ends exactly there is served (and equals paged_decode row by row, which splits the
    context instead of staging its table); a flash-sized segment has no bound."""
    dtype, D, nq = case
    BS = 16
    lib = ops._lib.load()
    code = {torch.float32: 0, torch.bfloat16: 1}[dtype]
    limit = lib.mi355_paged_prefill_max_context(code, D, BS, nq)
    assert 100000 < limit <= 8190 * BS and lib.mi355_paged_prefill_max_context(code, D, BS, 64 if D == 64 else 8) in (limit, 2 ** 31 - 1)
    assert lib.mi355_paged_prefill_max_context(code, 48, BS, nq) == 0
    rng = np.random.default_rng(D)
    per = limit // BS
    g = torch.Generator(device=device).manual_seed(D)
    cache = (torch.randn((2, per + 2, 1, BS, D), generator=g, device=device) * 0.5).to(dtype)
    bt = torch.from_numpy(rng.permutation(per + 2)[:per].astype(np.int32)).view(1, per).to(device)
    q = (torch.randn((nq, 1, D), generator=g, device=device)).to(dtype)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    got = ops.paged_prefill(q, cache[0], cache[1], bt, i32([0]), i32([nq]), i32([limit - nq]), nq, limit)
    rows = ops.paged_decode(q, cache[0], cache[1], bt.expand(nq, per).contiguous(),
                            i32(list(range(limit - nq + 1, limit + 1))), limit)
    torch.cuda.synchronize()
    tol = dict(atol=2e-6, rtol=1e-5) if dtype == torch.float32 else dict(atol=4e-3, rtol=2e-2)
    np.testing.assert_allclose(got.float().cpu().numpy(), rows.float().cpu().numpy(), **tol)


@pytest.mark.parametrize('shape', [
    # dtype, Hq, Hkv, D, BS, [(pos0, n_query_tokens), ...]
    (torch.float32, 4, 4, 32, 16, [(0, 40), (130, 17), (5, 1), (64, 64)]),    # the reference engine's default geometry
    (torch.float32, 8, 2, 64, 8, [(1000, 21), (3, 9)]),                       # grouped heads, long prefix
    (torch.float32, 6, 6, 128, 3, [(10, 29)]),                                # block size not a power of two
    (torch.float32, 8, 2, 64, 16, [(100, 200), (0, 70), (7, 3)]),             # long chunks
    (torch.float32, 4, 4, 128, 8, [(31, 100)]),
    (torch.float32, 4, 2, 32, 3, [(50, 129), (0, 64)]),                       # block size not a power of two
    (torch.bfloat16, 4, 4, 32, 16, [(33, 50), (0, 7)]),                       # 16-bit at a head_dim off the matrix-core path
    (torch.float16, 2, 1, 256, 4, [(20, 30)]),
])
@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
def test_paged_prefill_segments_on_the_valu_kernel(ops, device, shape, layout):
    """fp32 caches and head dims the matrix-core kernels do not take: chunks of new tokens over the
    paged cache run the VALU kernel with eight consecutive query rows per wave; checked token by
    token against the decode oracle, rows outside every segment untouched."""
    dtype, Hq, Hkv, D, BS, segs = shape
    rng = np.random.default_rng(Hq * 1000 + D + BS)
    nseg = len(segs)
    max_ctx = max(p0 + n for p0, n in segs)
    per = (max_ctx + BS - 1) // BS
    NB = nseg * per + 5
    bt = rng.permutation(NB)[:nseg * per].reshape(nseg, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(dtype)
    T = sum(n for _, n in segs) + 3
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    q0, starts = 2, []
    for _, n in segs:
        starts.append(q0)
        q0 += n
    tc = cache.to(device)
    if layout == 'NSHD':
        tc = tc.permute(0, 1, 3, 2, 4).contiguous()
    out = torch.full((T, Hq, D), 7.0, dtype=dtype, device=device)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
    ops.paged_prefill(q.to(device), tc[0], tc[1], torch.from_numpy(bt).to(device), i32(starts),
                      i32([n for _, n in segs]), i32([p0 for p0, _ in segs]), max(n for _, n in segs), max_ctx,
                      layout=layout, out=out)
    got = out.float().cpu().numpy()
    qn, cn = q.float().numpy(), cache.float().numpy()
    owned = np.zeros(T, dtype=bool)
    for s, (p0, n) in enumerate(segs):
        for j in range(n):
            row = starts[s] + j
            owned[row] = True
            want = orc.paged_decode(qn[row], cn[0], cn[1], bt[s].tolist(), p0 + j + 1, BS, dtype=np.float64)
            np.testing.assert_allclose(got[row], want, **(F32_TOL if dtype == torch.float32 else LOWP_TOL))
    assert (got[~owned] == 7.0).all()


def test_decode_ignores_unused_and_clamps_bad_block_ids(ops, device):
    """Entries of a block-table row past ceil(context/block_size) are never read, whatever they
    hold; ids outside the pool inside the used range are clamped (wrong answer for that row,
    but no out-of-bounds access and no effect on other rows)."""
    rng = np.random.default_rng(77)
    q, cache, bt, ctx = _random_batch(rng, 4, 8, 8, 64, 16, 160, ragged=False)
    ctx = [160, 33, 16, 97]
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, 16, dtype=np.float64)
    poisoned = bt.copy()
    for b, n in enumerate(ctx):
        poisoned[b, (n + 15) // 16:] = [-7, 2 ** 31 - 1, -2 ** 31, 123456789][b]
    out = _decode_on_gpu(ops, device, q, cache, poisoned, ctx, torch.float32)
    np.testing.assert_allclose(out, want, **F32_TOL)
    bad = bt.copy()
    bad[2, 0] = 2 ** 30                                 # inside the used range of row 2 only
    out = _decode_on_gpu(ops, device, q, cache, bad, ctx, torch.float32)
    np.testing.assert_allclose(out[[0, 1, 3]], want[[0, 1, 3]], **F32_TOL)
    assert np.isfinite(out[2]).all()


def test_decode_zero_context_rows(ops, device):
    rng = np.random.default_rng(5)
    q, cache, bt, ctx = _random_batch(rng, 4, 8, 8, 64, 16, 64)
    ctx[1] = 0
    ctx[3] = 0
    out = _decode_on_gpu(ops, device, q, cache, bt, ctx, torch.float32)
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, 16)
    np.testing.assert_allclose(out, want, **F32_TOL)
    assert not out[1].any() and not out[3].any()


def test_decode_softmax_spike(ops, device):
    """A late key that dominates every earlier one forces the running-max rescale path."""
    rng = np.random.default_rng(11)
    q, cache, bt, ctx = _random_batch(rng, 2, 8, 8, 64, 16, 600, ragged=False)
    for b in range(2):
        pos = 500 + b
        blk, row = bt[b][pos // 16], pos % 16
        cache[0, blk, :, row, :] = q[b] * 4.0     # score ~ 4*|q|^2/8 >> the rest
    want = orc.paged_decode_batch(q, cache[0], cache[1], bt, ctx, 16, dtype=np.float64)
    out = _decode_on_gpu(ops, device, q, cache, bt, ctx, torch.float32)
    np.testing.assert_allclose(out, want, atol=2e-5, rtol=1e-4)


def test_decode_metric_shape_subset_bf16(ops, device):
    """The benchmark configuration (B=256, S=2048, 8 heads, D=64, BS=16, bf16), checked on a
    random subset of sequences against the float64 oracle, plus two size-independent
    properties: rows are independent of batch position and of physical block placement."""
    B, S, H, D, BS = 256, 2048, 8, 64, 16
    g = torch.Generator(device='cpu').manual_seed(0)
    NB = B * S // BS + S // BS
    kc = torch.randn((NB, H, BS, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    vc = torch.randn((NB, H, BS, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((B, H, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    bt = torch.randperm(NB, generator=g)[:B * S // BS].view(B, S // BS).to(torch.int32)
    ctx = torch.full((B,), S, dtype=torch.int32)
    out = ops.paged_decode(q.to(device), kc.to(device), vc.to(device), bt.to(device), ctx.to(device), S)
    torch.cuda.synchronize()
    out = out.float().cpu().numpy()
    kcf, vcf = kc.float().numpy(), vc.float().numpy()
    for b in (0, 17, 101, 255):
        want = orc.paged_decode(q[b].float().numpy(), kcf, vcf, bt[b].tolist(), S, BS, dtype=np.float64)
        np.testing.assert_allclose(out[b], want, **LOWP_TOL)
    # property 1: a reversed batch gives reversed rows, bit for bit
    rev = torch.arange(B - 1, -1, -1)
    out_rev = ops.paged_decode(q[rev].to(device), kc.to(device), vc.to(device), bt[rev].to(device),
                               ctx.to(device), S)
    assert torch.equal(out_rev.cpu(), torch.from_numpy(out).to(torch.bfloat16)[rev])
    # property 2: moving a sequence's blocks elsewhere in the pool changes nothing
    used = set(bt.flatten().tolist())
    spare = torch.tensor([i for i in range(NB) if i not in used][:S // BS])
    kc2, vc2, bt2 = kc.clone(), vc.clone(), bt.clone()
    kc2[spare] = kc[bt[3].long()]
    vc2[spare] = vc[bt[3].long()]
    bt2[3] = spare.to(torch.int32)
    out2 = ops.paged_decode(q[3:4].to(device), kc2.to(device), vc2.to(device), bt2[3:4].to(device),
                            ctx[3:4].to(device), S)
    np.testing.assert_allclose(out2.float().cpu().numpy()[0], out[3], **LOWP_TOL)


@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
def test_decode_config2_all_rows_bf16(ops, device, layout):
    """The single-GPU configuration B=64, S=1024, 8 heads, D=64, BS=16, bf16, scattered block table.
    Its launch plan differs from the benchmark shape's (fewer kv heads per workgroup, the freed
    waves take token sub-ranges of a head and merge through LDS), so the exact workload is checked:
    every row against the float64 oracle on the rounded inputs, plus batch-order equivariance
    (bit for bit) and invariance under moving a sequence's blocks, in both cache layouts."""
    B, S, H, D, BS = 64, 1024, 8, 64, 16
    per = S // BS
    g = torch.Generator(device='cpu').manual_seed(2)
    NB = B * per + 64
    kc = torch.randn((NB, H, BS, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    vc = torch.randn((NB, H, BS, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((B, H, D), generator=g, dtype=torch.float32).to(torch.bfloat16)
    bt = torch.randperm(NB, generator=g)[:B * per].view(B, per).to(torch.int32)
    ctx = torch.full((B,), S, dtype=torch.int32)

    def dev_cache(t):
        t = t.to(device)
        return t.permute(0, 2, 1, 3).contiguous() if layout == 'NSHD' else t

    kd, vd = dev_cache(kc), dev_cache(vc)
    out_t = ops.paged_decode(q.to(device), kd, vd, bt.to(device), ctx.to(device), S, layout=layout)
    torch.cuda.synchronize()
    out = out_t.float().cpu().numpy()
    want = orc.paged_decode_batch(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), bt.numpy(),
                                  [S] * B, BS, dtype=np.float64)
    np.testing.assert_allclose(out, want, **LOWP_TOL)
    # a permuted batch gives permuted rows, bit for bit
    perm = torch.randperm(B, generator=g)
    out_perm = ops.paged_decode(q[perm].to(device), kd, vd, bt[perm].to(device), ctx.to(device), S, layout=layout)
    assert torch.equal(out_perm.cpu(), out_t.cpu()[perm])
    # moving the blocks of sequence 5 into the spare blocks of the pool changes nothing, bit for bit
    used = set(bt.flatten().tolist())
    spare = torch.tensor([i for i in range(NB) if i not in used][:per])
    kc2, vc2, bt2 = kc.clone(), vc.clone(), bt.clone()
    kc2[spare] = kc[bt[5].long()]
    vc2[spare] = vc[bt[5].long()]
    bt2[5] = spare.to(torch.int32)
    out2 = ops.paged_decode(q.to(device), dev_cache(kc2), dev_cache(vc2), bt2.to(device), ctx.to(device), S,
                            layout=layout)
    assert torch.equal(out2.cpu(), out_t.cpu())
    # ragged contexts at the same batch and bound: rows again equal the oracle
    rctx = torch.randint(1, S + 1, (B,), generator=g, dtype=torch.int32)
    out3 = ops.paged_decode(q.to(device), kd, vd, bt.to(device), rctx.to(device), S, layout=layout)
    want3 = orc.paged_decode_batch(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), bt.numpy(),
                                   rctx.tolist(), BS, dtype=np.float64)
    np.testing.assert_allclose(out3.float().cpu().numpy(), want3, **LOWP_TOL)


def test_decode_grouped_query_large_shape_properties(ops, device):
    """A serving-size grouped-query batch (64 sequences x 4096 tokens, 32 query / 8 kv heads,
    D=128, bf16; 1 GiB of KV) through the matrix-core kernel: a subset of rows against the
    float64 oracle, and the size-independent properties -- batch order does not matter (bit for
    bit), a longer block table / larger max_context bound does not change a row, and the split
    path (forced by a small batch) agrees with the unsplit one within tolerance."""
    B, S, Hq, Hkv, D, BS = 64, 4096, 32, 8, 128, 16
    g = torch.Generator(device=device).manual_seed(3)
    per = S // BS
    NB = B * per + 11
    kc = torch.randn((NB, Hkv, BS, D), generator=g, device=device).to(torch.bfloat16)
    vc = torch.randn((NB, Hkv, BS, D), generator=g, device=device).to(torch.bfloat16)
    q = torch.randn((B, Hq, D), generator=g, device=device).to(torch.bfloat16)
    bt = torch.randperm(NB, generator=g, device=device)[:B * per].view(B, per).to(torch.int32)
    ctx = torch.randint(S // 2, S + 1, (B,), generator=g, device=device, dtype=torch.int32)
    ctx[0] = S
    out = ops.paged_decode(q, kc, vc, bt, ctx, S)
    kcf, vcf = kc.float().cpu().numpy(), vc.float().cpu().numpy()
    for b in (0, 9, 63):
        want = orc.paged_decode(q[b].float().cpu().numpy(), kcf, vcf, bt[b].tolist(), int(ctx[b]), BS, dtype=np.float64)
        np.testing.assert_allclose(out[b].float().cpu().numpy(), want, **LOWP_TOL)
    perm = torch.randperm(B, generator=g, device=device)
    assert torch.equal(ops.paged_decode(q[perm], kc, vc, bt[perm], ctx[perm], S), out[perm])
    wide = torch.cat([bt, torch.zeros((B, 40), dtype=torch.int32, device=device)], dim=1)
    assert torch.equal(ops.paged_decode(q, kc, vc, wide, ctx, S), out)
    few = ops.paged_decode(q[:3], kc, vc, bt[:3], ctx[:3], S)            # 3 sequences: tokens are split
    torch.testing.assert_close(few.float(), out[:3].float(), **LOWP_TOL)


def test_result_tensors_supplied_by_the_caller_are_checked(ops, device):
    """`out=` must be the dense tensor the kernel writes; a view with other strides, another dtype or
    another shape is refused instead of being written in the wrong places."""
    rng = np.random.default_rng(3)
    q, cache, bt, ctx = _random_batch(rng, 3, 8, 8, 64, 16, 40)
    tq = torch.from_numpy(q).to(device)
    tc = torch.from_numpy(cache).to(device)
    tb = torch.from_numpy(bt).to(device)
    cl = torch.tensor(ctx, dtype=torch.int32, device=device)
    good = torch.empty_like(tq)
    assert ops.paged_decode(tq, tc[0], tc[1], tb, cl, 40, out=good) is good
    wide = torch.empty((3, 8, 128), device=device)
    for bad in (wide[:, :, :64], torch.empty((3, 8, 64), dtype=torch.float16, device=device),
                torch.empty((4, 8, 64), device=device), torch.empty((3, 8, 64))):
        with pytest.raises(ValueError):
            ops.paged_decode(tq, tc[0], tc[1], tb, cl, 40, out=bad)
    logits = torch.randn(4, 100, device=device)
    with pytest.raises(ValueError):
        ops.sample_argmax(logits, out=torch.empty(3, dtype=torch.int32, device=device))
    with pytest.raises(ValueError):
        ops.sample_argmax(logits, out=torch.empty(4, dtype=torch.int64, device=device))
    assert ops.sample_argmax(logits, out=torch.empty(6, dtype=torch.int32, device=device))[:4].tolist() == \
        logits.argmax(dim=1).tolist()


def test_decode_bad_arguments(ops, device):
    from mi355_paged._lib import Mi355Error
    q = torch.zeros((1, 8, 48), device=device)
    kc = torch.zeros((4, 8, 16, 48), device=device)
    bt = torch.zeros((1, 4), dtype=torch.int32, device=device)
    cl = torch.ones((1,), dtype=torch.int32, device=device)
    with pytest.raises(Mi355Error, match='head_dim'):
        ops.paged_decode(q, kc, kc, bt, cl, 1)
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.paged_decode(q.cpu(), kc, kc, bt, cl, 1)


# ------------------------------------------------------------------ KV write / prefill
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
def test_kv_write_slots_exact(ops, device, dtype, layout):
    rng = np.random.default_rng(3)
    NB, H, BS, D, T = 9, 4, 8, 32, 37
    cache = torch.from_numpy(rng.standard_normal((2, NB, H, BS, D), dtype=np.float32)).to(dtype)
    k = torch.from_numpy(rng.standard_normal((T, H, D), dtype=np.float32)).to(dtype)
    v = torch.from_numpy(rng.standard_normal((T, H, D), dtype=np.float32)).to(dtype)
    slots = rng.permutation(NB * BS)[:T].astype(np.int64)
    slots[5] = -1                      # padding slot: skipped
    want = cache.float().numpy().copy()
    keep = slots >= 0
    orc.kv_write_slots(want[0], want[1], k.float().numpy()[keep], v.float().numpy()[keep], slots[keep], BS)
    dc = cache.to(device)
    if layout == 'NSHD':
        dc = dc.permute(0, 1, 3, 2, 4).contiguous()
    ops.kv_write_slots(dc[0], dc[1], k.to(device), v.to(device), torch.from_numpy(slots).to(device), layout)
    torch.cuda.synchronize()
    got = dc.permute(0, 1, 3, 2, 4) if layout == 'NSHD' else dc
    np.testing.assert_array_equal(got.float().cpu().numpy(), want)


@pytest.mark.parametrize('case', npz_cases('g3_prefill_cases.npz'), ids=lambda c: str(c[0]))
def test_g3_prefill_cases_fp32(ops, device, case):
    import paged_attention as pa
    m, f = case
    cache = torch.from_numpy(f['cache_before'].astype(np.float32)).to(device)
    out = pa.paged_attention_prefill(torch.from_numpy(f['query'].astype(np.float32)).to(device),
                                     torch.from_numpy(f['key'].astype(np.float32)).to(device),
                                     torch.from_numpy(f['value'].astype(np.float32)).to(device),
                                     cache, f['block_table'].tolist(), m['bs'])
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), f['out'], **F32_TOL)
    np.testing.assert_array_equal(cache.cpu().numpy(), f['cache_after'].astype(np.float32))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_prefill_varlen_packed(ops, device, dtype):
    rng = np.random.default_rng(8)
    lens = [1, 17, 64, 130, 5]
    Hq, Hkv, D = 8, 4, 64
    T = sum(lens)
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    k = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    v = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = orc.varlen_causal_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu,
                                       dtype=np.float64)
    out = ops.prefill_causal(q.to(device), k.to(device), v.to(device), torch.from_numpy(cu).to(device),
                             max(lens))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), want,
                               **(F32_TOL if dtype == torch.float32 else LOWP_TOL))


@pytest.mark.parametrize('shape', [
    # Hq, Hkv, D, lens
    (4, 4, 32, [1, 7, 8, 9, 33, 250]),          # the reference engine's default geometry; lengths around the group of 8
    (8, 2, 64, [130, 5, 64]),                   # grouped heads: one workgroup per query head of the kv group
    (6, 6, 128, [17, 300]),                     # six kv heads: two per workgroup, waves split the keys
    (3, 1, 256, [40, 9]),
])
def test_prefill_fp32_row_groups_match_oracle_and_single_rows(ops, device, shape):
    """fp32 prompts: the fp32 matrix instruction at head_dim 32 / 64 / 128 (the default there), the
    vector-ALU kernel with one query row per lane, elsewhere the decode-shaped kernel with eight
    consecutive query rows per wave sharing each K/V load; all meet the oracle and the
    one-row-per-workgroup form of the decode-shaped kernel."""
    Hq, Hkv, D, lens = shape
    rng = np.random.default_rng(sum(lens) + D)
    T = sum(lens)
    q, k, v = (torch.from_numpy(rng.standard_normal((T, h, D), dtype=np.float32)) for h in (Hq, Hkv, Hkv))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = orc.varlen_causal_attention(q.numpy(), k.numpy(), v.numpy(), cu, dtype=np.float64)
    args = (q.to(device), k.to(device), v.to(device), torch.from_numpy(cu).to(device), max(lens))
    default = ops.prefill_causal(*args).cpu().numpy()          # head_dim 32 / 64 / 128: matrix cores
    with ops.debug_knob('MI355_PREFILL_F32_MFMA', 0):
        per_lane = ops.prefill_causal(*args).cpu().numpy()     # head_dim 32 / 64 / 128: one query row per lane
        with ops.debug_knob('MI355_PREFILL_F32_TILED', 0):
            grouped = ops.prefill_causal(*args).cpu().numpy()
            with ops.debug_knob('MI355_PREFILL_ROW_GROUPS', 0):
                single = ops.prefill_causal(*args).cpu().numpy()
    for got in (default, per_lane, grouped, single):
        np.testing.assert_allclose(got, want, **F32_TOL)
    np.testing.assert_allclose(grouped, single, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(per_lane, single, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(default, single, atol=4e-6, rtol=1e-5)


@pytest.mark.parametrize('D', [32, 64, 128])
def test_prefill_fp32_matrix_cores_long_and_growing_scores(ops, device, D):
    """The fp32 matrix-core kernel over several tiles per row (sequence lengths around the 128-row
    workgroup and 64-key tile borders, grouped heads, strided views), with scores that rise along
    the sequence by about 200 log2 units so that the deferred softmax reference moves many times;
    packed and paged entry points.  Tolerance: atol 5e-5 (the reference's 1e-5 bar is held on ordinary
    data by the golden and shape tests; here a row's scores span 200 log2 units and its weights pass
    through several moves of the reference, each an exp2 of about an ulp)."""
    rng = np.random.default_rng(D)
    Hq, Hkv, BS = 4, 2, 16
    lens = [600, 129, 64, 1, 257]
    T = sum(lens)
    q = rng.standard_normal((T, Hq, D), dtype=np.float32) * 0.2
    k = rng.standard_normal((T, Hkv, D), dtype=np.float32) * 0.2
    v = rng.standard_normal((T, Hkv, D), dtype=np.float32)
    q[:600, :, 0] = 4.0
    ramp = np.arange(600, dtype=np.float32) * 0.45
    ramp[300:] += 40.0
    ramp[437] += 30.0
    k[:600, :, 0] = ramp[:, None] * math.sqrt(D) / 4.0 * 0.25
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = orc.varlen_causal_attention(q, k, v, cu, dtype=np.float64)
    tq, tk, tv = (torch.from_numpy(a).to(device) for a in (q, k, v))
    got = ops.prefill_causal(tq, tk, tv, torch.from_numpy(cu).to(device), max(lens))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=5e-5, rtol=1e-4)
    # the same rows as chunks over a block cache: sequence s = a prefix of pre[s] cached tokens + the rest as queries
    pre = [300, 0, 17, 0, 100]
    per = (max(lens) + BS - 1) // BS
    NB = len(lens) * per + 3
    bt = rng.permutation(NB)[:len(lens) * per].reshape(len(lens), per).astype(np.int32)
    cache = np.zeros((2, NB, Hkv, BS, D), dtype=np.float32)
    for s, n in enumerate(lens):
        for t in range(n):
            cache[0, bt[s, t // BS], :, t % BS] = k[cu[s] + t]
            cache[1, bt[s, t // BS], :, t % BS] = v[cu[s] + t]
    tc = torch.from_numpy(cache).to(device)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=device)
    q0 = [int(cu[s]) + pre[s] for s in range(len(lens))]
    nq = [lens[s] - pre[s] for s in range(len(lens))]
    out = torch.zeros_like(tq)
    ops.paged_prefill(tq, tc[0], tc[1], torch.from_numpy(bt).to(device), i32(q0), i32(nq), i32(pre), max(nq), max(lens),
                      out=out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for s in range(len(lens)):
        a, b = q0[s], int(cu[s + 1])
        np.testing.assert_allclose(got[a:b], want[a:b], atol=5e-5, rtol=1e-4)
        assert not got[int(cu[s]):a].any()                   # rows outside every segment stay untouched


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('D', [64, 128])
def test_prefill_scores_that_keep_growing(ops, device, dtype, D):
    """The kernel moves its softmax reference only when a row's maximum outgrows it by 2^8.  Here
    the scores of every query rise along the sequence by about 200 log2 units in total (steps and
    jumps), so the reference moves many times, sometimes in the middle of a tile, and weights taken
    against an older reference come close to the deferral bound.  Packed and paged entry points."""
    rng = np.random.default_rng(D)
    T, Hq, Hkv, BS = 600, 4, 2, 16
    q = rng.standard_normal((T, Hq, D), dtype=np.float32) * 0.2
    k = rng.standard_normal((T, Hkv, D), dtype=np.float32) * 0.2
    v = rng.standard_normal((T, Hkv, D), dtype=np.float32)
    q[:, :, 0] = 4.0
    ramp = np.arange(T, dtype=np.float32) * 0.45                      # score ~ scale * 4 * ramp: rises steadily
    ramp[300:] += 40.0                                                # and jumps by more than the bound at once
    ramp[437] += 30.0                                                 # a single dominant key inside a tile
    k[:, :, 0] = (ramp * (D ** 0.5) / 16.0)[:, None]
    q, k, v = (torch.from_numpy(x).to(dtype) for x in (q, k, v))
    cu = np.array([0, T], dtype=np.int32)
    want = orc.varlen_causal_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, dtype=np.float64)
    out = ops.prefill_causal(q.to(device), k.to(device), v.to(device), torch.from_numpy(cu).to(device), T)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, **LOWP_TOL)
    # the same rows through the paged entry point: first 200 tokens cached, 400 new ones
    per = (T + BS - 1) // BS
    bt = rng.permutation(per + 2)[:per].astype(np.int32)
    kc = torch.zeros((per + 2, Hkv, BS, D), dtype=dtype)
    vc = torch.zeros((per + 2, Hkv, BS, D), dtype=dtype)
    for t in range(T):
        kc[bt[t // BS], :, t % BS] = k[t]
        vc[bt[t // BS], :, t % BS] = v[t]
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=device)
    got = ops.paged_prefill(q[200:].to(device), kc.to(device), vc.to(device), i32([bt.tolist()]), i32([0]), i32([400]),
                            i32([200]), 400, T)
    np.testing.assert_allclose(got.float().cpu().numpy(), want[200:], **LOWP_TOL)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # lens, Hq, Hkv, D
    ([128], 8, 8, 64), ([129, 1, 63, 64, 65], 8, 8, 64), ([700], 4, 2, 64), ([300, 257], 8, 1, 128),
    ([33, 500, 2], 2, 2, 128),
])
def test_prefill_matrix_core_kernel(ops, device, shape, dtype):
    """The MFMA prompt kernel (16-bit, head_dim 64/128) against the float64 oracle, and
    against the VALU kernel it replaces (same inputs, MI355_PREFILL_VALU=1)."""
    import os
    lens, Hq, Hkv, D = shape
    rng = np.random.default_rng(sum(lens) + D)
    T = sum(lens)
    q = torch.from_numpy(rng.standard_normal((T, Hq, D), dtype=np.float32)).to(dtype)
    k = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    v = torch.from_numpy(rng.standard_normal((T, Hkv, D), dtype=np.float32)).to(dtype)
    # a dominant late key in the first sequence forces the running-max rescale of O
    if lens[0] > 100:
        k[90, :, :] = (q[95, ::Hq // Hkv, :].float() * 3).to(dtype)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want = orc.varlen_causal_attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu,
                                       dtype=np.float64)
    args = (q.to(device), k.to(device), v.to(device), torch.from_numpy(cu).to(device), max(lens))
    out = ops.prefill_causal(*args)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), want, **LOWP_TOL)
    with ops.debug_knob('MI355_PREFILL_VALU', 1):
        ref = ops.prefill_causal(*args)
        torch.cuda.synchronize()
    np.testing.assert_allclose(ref.float().cpu().numpy(), want, **LOWP_TOL)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.float().cpu().numpy(), atol=2e-2, rtol=3e-2)


def test_prefill_strided_qkv_views(ops, device):
    """q/k/v as views into one fused projection output, the way the engine passes them."""
    rng = np.random.default_rng(2)
    T, H, D = 150, 8, 64
    qkv = torch.from_numpy(rng.standard_normal((T, 3 * H * D), dtype=np.float32)).to(torch.bfloat16).to(device)
    q, k, v = (t.reshape(T, H, D) for t in qkv.chunk(3, dim=-1))
    cu = torch.tensor([0, 100, 150], dtype=torch.int32, device=device)
    out = ops.prefill_causal(q, k, v, cu, 100)
    want = orc.varlen_causal_attention(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(),
                                       [0, 100, 150], dtype=np.float64)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, **LOWP_TOL)


def test_slot_mapping_device(ops, device):
    rng = np.random.default_rng(21)
    B, max_blocks, BS = 6, 12, 8
    bt = rng.integers(0, 500, size=(B, max_blocks)).astype(np.int32)
    starts = rng.integers(0, 40, size=B).astype(np.int32)
    counts = rng.integers(0, 50, size=B)
    counts = np.minimum(counts, max_blocks * BS - starts)
    cu = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    want = []
    for b in range(B):
        want += orc.slot_mapping_range(bt[b], int(starts[b]), int(counts[b]), BS)
    got = ops.slot_mapping(torch.from_numpy(bt).to(device), torch.from_numpy(starts).to(device),
                           torch.from_numpy(cu[:-1].copy()).to(device), int(cu[-1]), BS)
    assert got.cpu().tolist() == want


# ------------------------------------------------------------------ sampler
def _row(vals):
    conv = {None: float('nan'), 'inf': float('inf'), '-inf': float('-inf')}
    return [conv[v] if (v is None or isinstance(v, str)) else v for v in vals]


def test_argmax_golden_rows(ops, device):
    g = load_json('g7_sampler.json')
    rows = torch.tensor([_row(r) for r in g['greedy_rows']], dtype=torch.float32, device=device)
    assert ops.sample_argmax(rows).cpu().tolist() == g['greedy_ids']
    z = load_npz('g7_sampler_big.npz')
    got = ops.sample_argmax(torch.from_numpy(z['logits']).to(device)).cpu().tolist()
    assert got == z['greedy_ids'].tolist()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('vocab', [1, 3, 1000, 32000, 50257])
def test_argmax_random_exact(ops, device, dtype, vocab):
    g = torch.Generator().manual_seed(vocab)
    logits = torch.randn((33, vocab), generator=g).to(dtype)
    if vocab >= 1000:
        logits[4, 777] = float('nan')
        logits[4, 12] = float('nan')
        logits[7, :] = 1.5            # all ties -> index 0
        logits[9, 900] = logits[9].float().max().item() + 1
        logits[9, 901] = logits[9, 900]
    want = orc.argmax_rows(logits.float().numpy())
    assert ops.sample_argmax(logits.to(device)).cpu().tolist() == want
    sliced = torch.zeros((33, vocab + 5), dtype=dtype)          # strided rows, unaligned
    sliced[:, 1:vocab + 1] = logits
    assert ops.sample_argmax(sliced.to(device)[:, 1:vocab + 1]).cpu().tolist() == want


def test_nucleus_kept_set_matches_reference(ops, device):
    cases = load_json('g7_sampler.json')['nucleus']
    logits = torch.tensor([c['logits'] for c in cases], dtype=torch.float32, device=device)
    temp = torch.tensor([c['temperature'] for c in cases], device=device)
    top_p = torch.tensor([c['top_p'] for c in cases], device=device)
    ids, probs = ops.sample_topk_topp(logits, temp, top_p, None, seed=1, return_probs=True)
    probs = probs.cpu().numpy()
    for i, c in enumerate(cases):
        kept = sorted(np.nonzero(probs[i])[0].tolist())
        assert kept == sorted(c['kept_ids_sorted'])                                   # exact set
        np.testing.assert_allclose(probs[i][c['kept_ids_sorted']], c['kept_probs_sorted'],
                                   rtol=1e-4, atol=1e-6)
        assert int(ids[i]) in kept


def test_nucleus_kept_set_matches_reference_at_vocab_32000(ops, device):
    """Fixture rows generated by the reference's Sampler at the benchmark's vocabulary: the
    kernel must keep exactly the reference's tokens and give them the reference's renormalised
    probabilities.  A fifth row repeats row 0 with a top-k limit added; the reference has no top-k,
    so that row is held to the oracle's restatement (top-k, then the nucleus cut) -- for top-k,
    parity with the reference is unpinned by construction."""
    z = load_npz('g7_sampler_vocab32000.npz')
    n = len(z['temperature'])
    logits = torch.from_numpy(np.concatenate([z['logits'], z['logits'][:1]])).to(device)
    temp = torch.from_numpy(np.concatenate([z['temperature'], z['temperature'][:1]])).to(device)
    top_p = torch.from_numpy(np.concatenate([z['top_p'], z['top_p'][:1]])).to(device)
    top_k = torch.tensor([0] * n + [5000], dtype=torch.int32, device=device)
    ids, probs = ops.sample_topk_topp(logits, temp, top_p, top_k, seed=3, return_probs=True)
    probs, ids = probs.cpu().numpy(), ids.cpu().numpy()
    for i in range(n):
        want_ids = z[f'kept_ids_sorted_{i}']
        assert np.array_equal(np.nonzero(probs[i])[0], np.sort(want_ids))          # exact set
        np.testing.assert_allclose(probs[i][want_ids], z[f'kept_probs_sorted_{i}'], rtol=1e-4, atol=1e-9)
        assert probs[i][ids[i]] > 0
    want, kept = orc.nucleus_probs(z['logits'][0], float(z['temperature'][0]), float(z['top_p'][0]), top_k=5000)
    assert kept.sum() == 5000 and np.array_equal(np.nonzero(probs[n])[0], np.nonzero(kept)[0])
    np.testing.assert_allclose(probs[n], want, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('vocab', [50, 1000, 5000, 32000, 70000])
def test_sampling_filters_match_oracle(ops, device, vocab):
    rng = np.random.default_rng(vocab)
    B = 6
    logits = (rng.standard_normal((B, vocab)) * 3).astype(np.float32)
    temps = [1.0, 0.7, 1.3, 0.5, 1.0, 2.0]
    top_ps = [0.9, 0.5, 1.0, 0.95, 1.0, 0.3]
    top_ks = [0, 0, 40, 10, 0, 5]
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device),
                                      torch.tensor(top_ps, device=device),
                                      torch.tensor(top_ks, dtype=torch.int32, device=device),
                                      seed=123, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(B):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        # the kept set may differ by the boundary token when the running mass is within float
        # rounding of top_p; everywhere else it must be identical
        diff = np.nonzero(got_kept != kept)[0]
        assert len(diff) <= 1, f"row {b}: kept sets differ at {diff}"
        if len(diff) == 0:
            np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-7)
        assert got_kept[int(ids[b])]
        assert abs(probs[b].sum() - 1.0) < 1e-4


def test_sampling_gives_nan_and_minus_inf_logits_probability_zero(ops, device):
    """Masked (-inf) and NaN logits carry no probability: the rest of the row is the softmax of the
    finite entries, and neither kind is ever drawn."""
    rng = np.random.default_rng(5)
    vocab = 1000
    logits = rng.standard_normal((4, vocab)).astype(np.float32)
    logits[:, 10] = -np.inf
    logits[1, 20] = np.nan
    logits[2, 0:500] = -np.inf
    logits[3, 7] = np.nan
    temp = torch.ones(4, device=device)
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), temp, None, None, seed=3, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(4):
        finite = np.isfinite(logits[b])
        want = np.zeros(vocab)
        e = np.exp(logits[b][finite].astype(np.float64) - logits[b][finite].max())
        want[finite] = e / e.sum()
        np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-9)
        assert finite[int(ids[b])]
    draws = torch.stack([ops.sample_topk_topp(torch.from_numpy(logits).to(device), temp, None, None, seed=s)
                         for s in range(50)]).cpu().numpy()
    for b in range(4):
        assert np.isfinite(logits[b][draws[:, b]]).all()


def test_sampling_top_k_without_top_p_on_a_peaked_row(ops, device):
    """top_p = 1 means "no nucleus cut" (05_mini_vllm/mini_vllm.py:619), also when a top-k limit is set:
    on a peaked row the mass above the small entries rounds to 1.0f, and a `mass < top_p` test
    would drop them although fewer than k entries precede them."""
    rng = np.random.default_rng(77)
    vocab = 257
    logits = (rng.standard_normal((3, vocab)) * 8).astype(np.float32)
    temps, top_ps, top_ks = [0.7, 0.7, 0.7], [1.0, 1.0, 1.0], [1000, vocab, 200]
    _, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device),
                                    torch.tensor(top_ps, device=device),
                                    torch.tensor(top_ks, dtype=torch.int32, device=device), seed=5, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(3):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        assert int((probs[b] > 0).sum()) == min(top_ks[b], vocab) == int(kept.sum())
        np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-30)


@pytest.mark.parametrize('vocab', [64, 3000, 32000, 40000])
def test_sampling_ties_at_the_cut_keep_lowest_indices(ops, device, vocab):
    """Rows built so that the cut falls inside a run of EQUAL probabilities: of the tied tokens the
    lowest indices are kept (the stable descending order of the oracle), for top-p, top-k and
    both, including a row where one value has more copies than the kernel's candidate buffer."""
    rng = np.random.default_rng(vocab + 1)
    rows, temps, top_ps, top_ks = [], [], [], []

    def add(x, t, p, k):
        rows.append(x.astype(np.float32)); temps.append(t); top_ps.append(p); top_ks.append(k)

    base = (rng.standard_normal(vocab) * 2).astype(np.float32)
    x = base.copy()
    tied = rng.choice(vocab, size=min(12, vocab // 4), replace=False)
    x[tied] = np.float32(1.5)                                    # a dozen equal logits in the upper tail
    add(x, 1.0, 0.6, 0)
    add(x, 1.0, 1.0, int((x > 1.5).sum()) + 5)                    # top-k cuts through the ties
    add(x, 0.7, 0.8, int((x > 1.5).sum()) + 3)
    y = np.full(vocab, -3.0, dtype=np.float32)                    # thousands of copies of one value
    y[rng.choice(vocab, size=3, replace=False)] = 2.0
    add(y, 1.0, 0.9, 0)
    add(y, 1.0, 1.0, max(4, vocab // 2))
    add(np.zeros(vocab, dtype=np.float32), 1.0, 0.5, 0)           # uniform row: keep the first half
    logits = np.stack(rows)
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device),
                                      torch.tensor(top_ps, device=device),
                                      torch.tensor(top_ks, dtype=torch.int32, device=device),
                                      seed=5, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(len(rows)):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        diff = np.nonzero(got_kept != kept)[0]
        # The oracle's float32 running sum over thousands of equal terms carries rounding of its own
        # (the kernel evaluates mass_above + j * value instead), so the NUMBER of tied copies kept may
        # differ by a fraction of a percent; which copies are kept may not: always the lowest indices.
        cut_value = logits[b][diff[0]] if len(diff) else None
        n_tied = int((logits[b] == cut_value).sum()) if len(diff) else 0
        assert len(diff) <= max(1, n_tied // 500), f"row {b}: kept sets differ at {diff[:10]} ({got_kept.sum()} vs {kept.sum()})"
        if len(diff):
            assert (logits[b][diff] == cut_value).all()
            tied_idx = np.nonzero(logits[b] == cut_value)[0]
            n_kept = int(got_kept[tied_idx].sum())
            assert got_kept[tied_idx[:n_kept]].all() and not got_kept[tied_idx[n_kept:]].any()
        if len(diff) == 0:
            np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-7)
        assert got_kept[int(ids[b])]


@pytest.mark.parametrize('vocab', [1500, 20000, 32000])
def test_sampling_top_k_sizes_around_the_bounded_tally(ops, device, vocab):
    """Top-k limits on both sides of the size up to which the kernel tallies only the values above a
    lower bound of the k-th largest (k' = ceil(k / waves) <= 8 per wave): exactly k tokens stay, the
    largest ones, with and without a nucleus cut on top -- also on rows whose large values all sit
    in a few threads' elements, repeat many times, or are all equal."""
    rng = np.random.default_rng(vocab + 9)
    rows, temps, top_ps, top_ks = [], [], [], []
    spread = (rng.standard_normal(vocab) * 2).astype(np.float32)
    clustered = (rng.standard_normal(vocab) * 0.5).astype(np.float32)
    clustered[100:140] += 9.0                                       # the 40 largest values are neighbours in memory
    repeats = rng.integers(0, 6, size=vocab).astype(np.float32)      # six distinct values, thousands of copies each
    for k in (1, 2, 16, 17, 50, 128, 129, 400):
        for x, p in ((spread, 1.0), (clustered, 1.0), (repeats, 1.0), (spread, 0.8)):
            rows.append(x); temps.append(1.0 if p == 1.0 else 0.8); top_ps.append(p); top_ks.append(k)
    rows.append(np.zeros(vocab, dtype=np.float32)); temps.append(1.0); top_ps.append(1.0); top_ks.append(50)
    logits = np.stack(rows)
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device),
                                      torch.tensor(top_ps, device=device),
                                      torch.tensor(top_ks, dtype=torch.int32, device=device), seed=11, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(len(rows)):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        if top_ps[b] == 1.0:
            assert int(got_kept.sum()) == min(top_ks[b], vocab), f"row {b} (k={top_ks[b]}): {got_kept.sum()} tokens kept"
            assert np.array_equal(got_kept, kept), f"row {b} (k={top_ks[b]})"
            np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-9)
        else:
            assert len(np.nonzero(got_kept != kept)[0]) <= 1
        assert got_kept[int(ids[b])]


def test_sampling_distribution(ops, device):
    """Draws follow the filtered distribution (chi-square style bound on 20k draws)."""
    vocab, n = 16, 20000
    logits = torch.linspace(-2, 2, vocab).repeat(n, 1).to(device)
    temp = torch.full((n,), 0.8, device=device)
    top_p = torch.full((n,), 0.9, device=device)
    ids = ops.sample_topk_topp(logits, temp, top_p, None, seed=7).cpu().numpy()
    want, _ = orc.nucleus_probs(logits[0].cpu().numpy(), 0.8, 0.9)
    freq = np.bincount(ids, minlength=vocab) / n
    assert np.all(freq[want == 0] == 0)
    assert np.all(np.abs(freq - want) <= 4 * np.sqrt(want * (1 - want) / n) + 1e-3)
    again = ops.sample_topk_topp(logits, temp, top_p, None, seed=7).cpu().numpy()
    assert np.array_equal(ids, again)                      # same seed -> same draws
    other = ops.sample_topk_topp(logits, temp, top_p, None, seed=8).cpu().numpy()
    assert not np.array_equal(ids, other)


def test_sampling_top_k_on_streamed_rows_at_extreme_temperatures(ops, device):
    """Rows beyond the registers under a top-k limit (elements far below a bound of the k-th largest
    x / T are skipped unevaluated), at temperatures that make x / T small, ordinary and so large that
    neighbouring floats are further apart than the bound's margin: exactly the k largest stay."""
    rng = np.random.default_rng(12)
    vocab = 50000
    x = (rng.standard_normal(vocab) * 3).astype(np.float32)
    rows, temps, top_ks = [], [], []
    for t in (50.0, 1.0, 0.02, 1e-5):
        for k in (1, 7, 50, 128):
            rows.append(x); temps.append(t); top_ks.append(k)
    logits = np.stack(rows)
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device), None,
                                      torch.tensor(top_ks, dtype=torch.int32, device=device), seed=4, return_probs=True)
    probs = probs.cpu().numpy()
    order = np.argsort(-x, kind='stable')
    for b in range(len(rows)):
        got = set(np.nonzero(probs[b] > 0)[0].tolist())
        # at the coldest temperatures tokens underflow to probability 0 and carry nothing: of the k
        # largest, those clearly above the smallest float must be there, those clearly below it not
        weight = {int(i): np.exp((np.float64(x[i]) - np.float64(x[order[0]])) / temps[b]) for i in order[:top_ks[b]]}
        must = {i for i, w in weight.items() if w > 1e-36}
        may = {i for i, w in weight.items() if w > 1e-46}
        assert must <= got <= may, f"row {b} (T={temps[b]}, k={top_ks[b]}): {len(got)} kept, {len(must)}..{len(may)} expected"
        assert abs(probs[b].sum() - 1.0) < 1e-4 and int(ids[b]) in got


def test_sampling_top_k_on_streamed_rows_from_the_candidate_list(ops, device):
    """Rows beyond the registers under a top-k limit keep the elements at or above the bound in a
    list of at most one per thread: rows whose candidates fit (distinct values; 600 copies of the
    largest value) and rows where they cannot (3000 copies; a constant row) keep the same set -- of
    equal values the lowest indices -- and the draws follow the kept distribution."""
    rng = np.random.default_rng(77)
    vocab = 70001
    base = (rng.standard_normal(vocab) * 2).astype(np.float32)
    rows = [base.copy()]
    for copies in (600, 3000):
        x = base.copy()
        x[rng.choice(vocab, size=copies, replace=False)] = 9.0
        rows.append(x)
    rows.append(np.full(vocab, 0.25, dtype=np.float32))
    k = 50
    logits = torch.from_numpy(np.stack(rows)).to(device)
    temps = torch.ones(len(rows), device=device)
    top_ks = torch.full((len(rows),), k, dtype=torch.int32, device=device)
    ids, probs = ops.sample_topk_topp(logits, temps, None, top_ks, seed=9, return_probs=True)
    probs = probs.cpu().numpy()
    kept_sets = []
    for b, x in enumerate(rows):
        want = np.sort(np.argsort(-x, kind='stable')[:k])
        got = np.nonzero(probs[b] > 0)[0]
        assert np.array_equal(got, want), f"row {b}: {len(got)} kept"
        ref = np.exp(x[want].astype(np.float64) - x[want].max())
        np.testing.assert_allclose(probs[b][want], ref / ref.sum(), rtol=2e-5, atol=1e-7)
        assert int(ids[b]) in set(want.tolist())
        kept_sets.append(want)
    # the same seed gives the same tokens, call after call (the list's order of arrival must not show)
    again = ops.sample_topk_topp(logits, temps, None, top_ks, seed=9)
    assert torch.equal(ids, again)
    # drawn frequencies over 60 seeds x 64 copies of each row against the kept distribution
    reps = 64
    tiled = logits.repeat_interleave(reps, dim=0)
    t_temps = torch.ones(tiled.shape[0], device=device)
    t_ks = torch.full((tiled.shape[0],), k, dtype=torch.int32, device=device)
    counts = np.zeros((len(rows), vocab), dtype=np.int64)
    nseeds = 60
    for seed in range(nseeds):
        got = ops.sample_topk_topp(tiled, t_temps, None, t_ks, seed=1000 + seed).cpu().numpy().reshape(len(rows), reps)
        for b in range(len(rows)):
            np.add.at(counts[b], got[b], 1)
    n = nseeds * reps
    for b in range(len(rows)):
        assert counts[b].sum() == n and counts[b][kept_sets[b]].sum() == n
        expect = probs[b][kept_sets[b]].astype(np.float64) * n
        sigma = np.sqrt(np.maximum(expect * (1 - expect / n), 1.0))
        assert np.all(np.abs(counts[b][kept_sets[b]] - expect) < 5.5 * sigma), f"row {b}"


@pytest.mark.parametrize('vocab', [900, 12000, 32000])     # the three register-resident kernel shapes
def test_sampling_top_k_on_register_rows_from_the_candidate_list(ops, device, vocab):
    """Rows that live in registers take cut, tie rule, kept mass and draw of a top-k limit (with or
    without a nucleus cut) from a short list of the candidates at or above a lower bound of the k-th
    largest weight.  Rows whose candidates fit (distinct values; 300 copies of the largest value, of
    which the lowest indices stay) and rows that fall back to the walk over the row (more copies
    than the list holds; a constant row; fewer positive weights than k) keep the oracle's set, the
    kept probabilities are the oracle's, the same seed gives the same tokens, and the draws -- with
    no distribution asked for, the production form -- follow the kept distribution."""
    rng = np.random.default_rng(vocab)
    base = (rng.standard_normal(vocab) * 2).astype(np.float32)
    rows, top_ps, top_ks = [base.copy(), base.copy(), base.copy()], [1.0, 0.6, 1.0], [50, 50, 128 if vocab > 1024 else 30]
    for copies in (300, 700):
        x = base.copy()
        x[rng.choice(vocab, size=copies, replace=False)] = 9.0
        rows.append(x); top_ps.append(1.0); top_ks.append(20)
    rows.append(np.full(vocab, 0.25, dtype=np.float32)); top_ps.append(1.0); top_ks.append(7)
    peaked = np.full(vocab, -200.0, dtype=np.float32)      # five positive weights, the rest underflow
    peaked[[3, 17, 400, 401, vocab - 1]] = [5.0, 4.0, 4.0, 3.0, 1.0]
    rows.append(peaked); top_ps.append(1.0); top_ks.append(12)
    # the k-th largest value is the list's bound itself: sixteen equal maxima, one in each wave's share of the row
    # (a thread owns the elements of 16-byte pieces tid, tid + 1024, ...), and more of the same value just below
    # the list's reach would tie with them -- the row must come out as the walk over the whole row gives it
    tied = (rng.standard_normal(vocab) * 0.5 - 6.0).astype(np.float32)
    tied[[min(256 * w, vocab - 1 - w) for w in range(16)]] = 3.0
    rows.append(tied); top_ps.append(1.0); top_ks.append(8)
    nrow = len(rows)
    logits = torch.from_numpy(np.stack(rows)).to(device)
    temps = torch.ones(nrow, device=device)
    tp = torch.tensor(top_ps, device=device)
    tk = torch.tensor(top_ks, dtype=torch.int32, device=device)
    ids, probs = ops.sample_topk_topp(logits, temps, tp, tk, seed=9, return_probs=True)
    probs = probs.cpu().numpy()
    kept_sets = []
    for b in range(nrow):
        want, kept = orc.nucleus_probs(rows[b], 1.0, top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        positive = want > 0
        assert np.array_equal(got_kept, kept & positive), f"row {b}: {got_kept.sum()} kept, {int(kept.sum())} expected"
        np.testing.assert_allclose(probs[b], want, rtol=2e-5, atol=1e-7)
        assert got_kept[int(ids[b])]
        kept_sets.append(np.nonzero(got_kept)[0])
    plain = ops.sample_topk_topp(logits, temps, tp, tk, seed=9)
    assert torch.equal(ids, plain)                          # asking for the distribution does not change the draw
    reps, nseeds = 64, 40
    tiled = logits.repeat_interleave(reps, dim=0)
    t_temps = torch.ones(tiled.shape[0], device=device)
    counts = np.zeros((nrow, vocab), dtype=np.int64)
    for seed in range(nseeds):
        got = ops.sample_topk_topp(tiled, t_temps, tp.repeat_interleave(reps), tk.repeat_interleave(reps),
                                   seed=500 + seed).cpu().numpy().reshape(nrow, reps)
        for b in range(nrow):
            np.add.at(counts[b], got[b], 1)
    n = nseeds * reps
    for b in range(nrow):
        assert counts[b].sum() == n and counts[b][kept_sets[b]].sum() == n, f"row {b}: a token outside the kept set"
        expect = probs[b][kept_sets[b]].astype(np.float64) * n
        sigma = np.sqrt(np.maximum(expect * (1 - expect / n), 1.0))
        assert np.all(np.abs(counts[b][kept_sets[b]] - expect) < 5.5 * sigma), f"row {b}"


@pytest.mark.parametrize('vocab', [5000, 20000, 32000, 50000, 128256])   # (the last two: rows beyond the registers)
def test_sampling_nucleus_from_flat_to_peaked_rows(ops, device, vocab):
    """Kept sets against the oracle for nucleus cuts from 0.3 to 0.999 on rows from flat to very
    peaked (a few tokens carry most of the mass, as a model's logits do), alone and under a top-k
    limit, and on a row whose mass sits in one token."""
    rng = np.random.default_rng(vocab + 3)
    rows, temps, top_ps, top_ks = [], [], [], []
    for std in (1.0, 2.5, 4.0, 7.0):
        x = (rng.standard_normal(vocab) * std).astype(np.float32)
        for p in (0.3, 0.9, 0.99, 0.999):
            rows.append(x); temps.append(1.0); top_ps.append(p); top_ks.append(0)
        rows.append(x); temps.append(0.7); top_ps.append(0.95); top_ks.append(40)
    onehot = np.full(vocab, -20.0, dtype=np.float32)
    onehot[vocab // 3] = 5.0                                         # one token holds everything
    rows.append(onehot); temps.append(1.0); top_ps.append(0.9); top_ks.append(0)
    logits = np.stack(rows)
    ids, probs = ops.sample_topk_topp(torch.from_numpy(logits).to(device), torch.tensor(temps, device=device),
                                      torch.tensor(top_ps, device=device),
                                      torch.tensor(top_ks, dtype=torch.int32, device=device), seed=21, return_probs=True)
    probs = probs.cpu().numpy()
    for b in range(len(rows)):
        want, kept = orc.nucleus_probs(logits[b], temps[b], top_ps[b], top_ks[b])
        got_kept = probs[b] > 0
        diff = np.nonzero(got_kept != kept)[0]
        # Where the cut runs through the far tail (top_p 0.99 and beyond: thousands of tokens of 1e-8
        # each) the oracle's float32 running sum over thousands of terms near 1 is only good to a few
        # 1e-6, i.e. dozens of such tokens: the sets may differ by tokens that together carry less than
        # a hundred-thousandth; elsewhere by one token.
        full = np.exp(logits[b].astype(np.float64) / temps[b] - (logits[b].astype(np.float64) / temps[b]).max())
        full /= full.sum()
        # (a float32 running sum of n terms wanders by about sqrt(n) half-units in the last place: with
        # 90 000 kept tokens of a 128256-entry row that is 2e-5 by itself)
        loose = max(1e-5, 4.0 * np.sqrt(float(kept.sum())) * 2.0 ** -24)
        assert len(diff) <= 1 or full[diff].sum() < loose, \
            f"row {b} (top_p {top_ps[b]}): kept sets differ at {diff[:8]} ({got_kept.sum()} vs {kept.sum()})"
        # The kernel sums the masses exactly (fixed point), so against the same rule in float64 -- a
        # token stays while the mass strictly above it is below top_p, at most top_k of them -- the
        # sets differ by no more than the rounding of the float32 probabilities and of their sum.
        order = np.argsort(-full, kind='stable')
        above = np.cumsum(full[order]) - full[order]
        stays = above < top_ps[b]
        if top_ks[b] > 0:
            stays[top_ks[b]:] = False
        stays[0] = True
        exact = np.zeros(vocab, dtype=bool)
        exact[order[stays]] = True
        off = np.nonzero(got_kept != exact)[0]
        assert len(off) <= 1 or full[off].sum() < 2e-6, \
            f"row {b} (top_p {top_ps[b]}): {len(off)} tokens of mass {full[off].sum():.3g} off the float64 cut"
        if len(diff) == 0:
            np.testing.assert_allclose(probs[b], want, rtol=2e-4, atol=1e-9)
        assert got_kept[int(ids[b])]


@pytest.mark.parametrize('dtype,vocab', [(torch.float32, 40000), (torch.float32, 70001), (torch.bfloat16, 131072)])
def test_sampling_streamed_rows_without_a_cut_follow_the_softmax(ops, device, dtype, vocab):
    """Rows longer than the registers hold, no cut, no distribution asked for (one trip through
    memory): 10 tokens carry the mass, at positions spread over the row; drawn frequencies must match
    their softmax at the given temperature, and masked (-inf) or NaN entries are never drawn."""
    rng = np.random.default_rng(vocab)
    B, calls, T = 256, 24, 0.8
    heavy = rng.choice(vocab, size=10, replace=False)
    heavy[0], heavy[1] = 0, vocab - 1                          # the row's first and last element among them
    x = np.full(vocab, -25.0, dtype=np.float32)
    x[heavy] = np.linspace(1.0, 3.0, 10).astype(np.float32)
    dead = rng.choice(np.setdiff1d(np.arange(vocab), heavy), size=50, replace=False)
    x[dead[:25]] = -np.inf
    x[dead[25:]] = np.nan
    logits = torch.from_numpy(np.tile(x, (B, 1))).to(device=device, dtype=dtype)
    xr = logits[0].float().cpu().numpy().astype(np.float64)
    ok = np.isfinite(xr)
    want = np.zeros(vocab)
    want[ok] = np.exp(xr[ok] / T - (xr[ok] / T).max())
    want /= want.sum()
    temp = torch.full((B,), T, device=device)
    draws = torch.stack([ops.sample_topk_topp(logits, temp, None, None, seed=99, offset=c * B) for c in range(calls)])
    draws = draws.cpu().numpy().ravel()
    assert ok[draws].all()
    n = draws.size
    for t in heavy:
        expect, got = want[t] * n, int((draws == t).sum())
        assert abs(got - expect) <= 5.0 * np.sqrt(expect) + 3, f"token {t}: {got} draws, {expect:.0f} expected"
    assert (np.isin(draws, heavy)).mean() > 0.999
    # a row with nothing finite has nothing to draw from: token 0, and no fault
    empty = torch.full((2, vocab), float('-inf'), device=device, dtype=dtype)
    assert ops.sample_topk_topp(empty, torch.ones(2, device=device), None, None, seed=1).cpu().tolist() == [0, 0]


def test_sampling_temperature_spread(ops, device):
    """Mirror of the reference's test_temperature_affects_distribution
    (05_mini_vllm/test_mini_vllm.py:195-211) on the kernel itself."""
    torch.manual_seed(42)
    logits = torch.randn(1, 100).repeat(200, 1).to(device)
    low = ops.sample_topk_topp(logits, torch.full((200,), 0.1, device=device), None, None, seed=1)
    high = ops.sample_topk_topp(logits, torch.full((200,), 10.0, device=device), None, None, seed=1)
    assert len(set(low.cpu().tolist())) < len(set(high.cpu().tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('B,Hq,Hkv,D,max_ctx', [(1, 8, 8, 64, 8192), (3, 8, 8, 64, 20000), (16, 8, 8, 128, 4096),
                                                (2, 4, 2, 64, 6000), (5, 2, 1, 32, 3000), (1, 8, 8, 64, 70000),
                                                (3, 32, 8, 128, 9000), (2, 16, 2, 64, 5000), (4, 40, 8, 128, 3000)])
def test_token_splits_merged_in_the_launch_match_the_second_launch(ops, device, dtype, B, Hq, Hkv, D, max_ctx):
    """The decode kernels (VALU, and the matrix-core one for grouped-query 16-bit shapes) merge their
    token splits themselves (the last workgroup of a group to finish combines the partials) -- against the two-launch form (MI355_DECODE_FUSED_MERGE=0), call after
    call on one workspace whose contents are scrambled in between (zeros, all-ones, random bits: the
    arrival cells must not depend on what the workspace held), ragged contexts including an empty
    sequence, and replays of one captured launch."""
    bs = 16
    g = torch.Generator(device=device).manual_seed(B * 131 + D)
    per = -(-max_ctx // bs)
    nb = B * per + 3
    kc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(dtype)
    vc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(dtype)
    bt = torch.randperm(nb, generator=g, device=device)[:B * per].view(B, per).to(torch.int32).contiguous()
    need = ops._lib.load().mi355_paged_decode_workspace_bytes(B, Hq, Hkv, D, bs, max_ctx, ops._DTYPE_CODE[dtype])
    assert need > 0, "the case is meant to split tokens across workgroups"
    tol = dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    stream = torch.cuda.current_stream(device).cuda_stream
    for it in range(12):
        q = torch.randn((B, Hq, D), generator=g, device=device).to(dtype)
        ctx = torch.randint(1, max_ctx + 1, (B,), generator=g, device=device).to(torch.int32)
        ctx[it % B] = max_ctx if it % 3 else 0
        with ops.debug_knob('MI355_DECODE_FUSED_MERGE', 0):
            want = ops.paged_decode(q, kc, vc, bt, ctx, max_ctx)
        ws, _ = ops._workspace(device, need, stream)
        if it % 4 == 0:
            ws.zero_()
        elif it % 4 == 1:
            ws.fill_(255)
        elif it % 4 == 2:
            ws.view(torch.int64)[:need // 8].random_(-2 ** 62, 2 ** 62, generator=g)
        with ops.debug_knob('MI355_DECODE_FUSED_MERGE', 65535):      # (by default only up to 8 splits)
            got = ops.paged_decode(q, kc, vc, bt, ctx, max_ctx)
        torch.testing.assert_close(got.float(), want.float(), **tol)
        assert not got[it % B].float().abs().sum().item() if not it % 3 else True

    q = torch.randn((B, Hq, D), generator=g, device=device).to(dtype)
    ctx = torch.full((B,), max_ctx, dtype=torch.int32, device=device)
    out = torch.empty_like(q)
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side), ops.debug_knob('MI355_DECODE_FUSED_MERGE', 65535):
        ops.paged_decode(q, kc, vc, bt, ctx, max_ctx, out=out)
    torch.cuda.current_stream(device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), ops.debug_knob('MI355_DECODE_FUSED_MERGE', 65535):
        ops.paged_decode(q, kc, vc, bt, ctx, max_ctx, out=out)
    for it in range(6):
        q.copy_(torch.randn((B, Hq, D), generator=g, device=device).to(dtype))
        ctx.copy_(torch.randint(1, max_ctx + 1, (B,), generator=g, device=device).to(torch.int32))
        graph.replay()
        with ops.debug_knob('MI355_DECODE_FUSED_MERGE', 0):
            want = ops.paged_decode(q, kc, vc, bt, ctx, max_ctx)
        torch.testing.assert_close(out.float(), want.float(), **tol)


@pytest.mark.gpu
@pytest.mark.parametrize('Hq,Hkv,D,B,S', [(8, 8, 64, 16, 8192), (32, 8, 128, 8, 8192), (8, 8, 64, 48, 3000)])
def test_interleaved_layout_keeps_heads_together_for_large_launches(ops, device, Hq, Hkv, D, B, S):
    """[block][row][head][d] caches: from 256 MiB of K/V on, the plan keeps enough kv heads in a
    workgroup for 512-byte pieces per token and fills the chip with token splits instead (merged in
    the launch).  Same numbers as the head-major layout gives for the same contents, ragged contexts."""
    bs = 16
    g = torch.Generator(device=device).manual_seed(Hq + D + B)
    per = -(-S // bs)
    nb = B * per + 3
    kc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(torch.bfloat16)
    vc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(torch.bfloat16)
    bt = torch.randperm(nb, generator=g, device=device)[:B * per].view(B, per).to(torch.int32).contiguous()
    q = torch.randn((B, Hq, D), generator=g, device=device).to(torch.bfloat16)
    ctx = torch.randint(1, S + 1, (B,), generator=g, device=device).to(torch.int32)
    ctx[0], ctx[1] = S, 0
    want = ops.paged_decode(q, kc, vc, bt, ctx, S)
    kc2, vc2 = kc.permute(0, 2, 1, 3).contiguous(), vc.permute(0, 2, 1, 3).contiguous()
    for _ in range(3):
        got = ops.paged_decode(q, kc2, vc2, bt, ctx, S, layout='NSHD')
        torch.testing.assert_close(got.float(), want.float(), rtol=1.6e-2, atol=2e-3)
    with ops.debug_knob('MI355_DECODE_FUSED_MERGE', 0):
        two = ops.paged_decode(q, kc2, vc2, bt, ctx, S, layout='NSHD')
    torch.testing.assert_close(two.float(), want.float(), rtol=1.6e-2, atol=2e-3)
    assert not got[1].float().abs().sum().item()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,D,Hq,Hkv', [(torch.bfloat16, 64, 8, 8), (torch.float32, 64, 8, 8), (torch.bfloat16, 128, 32, 8),
                                            (torch.float16, 128, 4, 4)])
def test_heads_spread_over_workgroups_match_the_plain_plan(ops, device, dtype, D, Hq, Hkv):
    """A launch without token splits over a cache with contiguous (block, head) tiles spreads its
    kv heads over up to MI355_PAGED_BALANCE_WGS workgroups (ragged batches finish together); the
    result equals the plain plan's (8 heads per workgroup) up to the order of the sub-range merge,
    for ragged contexts including empty and full ones, and the interleaved layout keeps the plain plan."""
    bs, B, S = 16, 128, 2048
    g = torch.Generator(device=device).manual_seed(D + Hq)
    per = S // bs
    nb = B * per + 5
    kc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(dtype)
    vc = torch.randn((nb, Hkv, bs, D), generator=g, device=device).to(dtype)
    bt = torch.randperm(nb, generator=g, device=device)[:B * per].view(B, per).to(torch.int32).contiguous()
    q = torch.randn((B, Hq, D), generator=g, device=device).to(dtype)
    ctx = torch.randint(1, S + 1, (B,), generator=g, device=device).to(torch.int32)
    ctx[0], ctx[1], ctx[2] = 0, S, 1
    assert ops._lib.load().mi355_paged_decode_workspace_bytes(B, Hq, Hkv, D, bs, S, ops._DTYPE_CODE[dtype]) == 0
    with ops.debug_knob('MI355_PAGED_BALANCE_WGS', 0):
        plain = ops.paged_decode(q, kc, vc, bt, ctx, S)
    spread = ops.paged_decode(q, kc, vc, bt, ctx, S)
    tol = dict(rtol=1e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=2e-3)
    torch.testing.assert_close(spread.float(), plain.float(), **tol)
    assert not spread[0].float().abs().sum().item()
    # [block][row][head][d]: one head per workgroup would read 128-byte pieces, so the plan stays
    kc2, vc2 = kc.permute(0, 2, 1, 3).contiguous(), vc.permute(0, 2, 1, 3).contiguous()
    a = ops.paged_decode(q, kc2, vc2, bt, ctx, S, layout='NSHD')
    with ops.debug_knob('MI355_PAGED_BALANCE_WGS', 0):
        b = ops.paged_decode(q, kc2, vc2, bt, ctx, S, layout='NSHD')
    assert torch.equal(a, b)
    torch.testing.assert_close(a.float(), plain.float(), **tol)
