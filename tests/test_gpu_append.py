"""mi355_paged_decode_append: cache write and attention of a decode step in one call (needs an MI355X).

The bar is bit-equality, output and caches, with the two calls it replaces -- mi355_kv_write_slots
at every sequence's last position followed by mi355_paged_decode -- which the other kernel tests
hold against the oracle and the golden vectors.  Before the fused call the slots to be written hold
NaN, so a read that did not wait for the write would show in the output.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from mi355_paged import ops as _ops
    return _ops


def _problem(rng, device, dtype, B, Hq, Hkv, D, BS, max_ctx, layout, ctx=None, spare=5):
    ctx = np.asarray(ctx if ctx is not None else rng.integers(1, max_ctx + 1, size=B), dtype=np.int64)
    per = (max_ctx + BS - 1) // BS
    NB = B * per + spare
    bt = rng.permutation(NB)[:B * per].reshape(B, per).astype(np.int32)
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(device=device, dtype=dtype)
    if layout == 'NSHD':
        cache = cache.permute(0, 1, 3, 2, 4).contiguous()
    q = torch.from_numpy(rng.standard_normal((B, Hq, D), dtype=np.float32)).to(device=device, dtype=dtype)
    kn = torch.from_numpy(rng.standard_normal((B, Hkv, D), dtype=np.float32)).to(device=device, dtype=dtype)
    vn = torch.from_numpy(rng.standard_normal((B, Hkv, D), dtype=np.float32)).to(device=device, dtype=dtype)
    last = np.maximum(ctx - 1, 0)
    slots = bt[np.arange(B), last // BS].astype(np.int64) * BS + last % BS
    slots[ctx < 1] = -1
    return dict(q=q, kn=kn, vn=vn, cache=cache, bt=torch.from_numpy(bt).to(device),
                ctx=torch.from_numpy(ctx.astype(np.int32)).to(device), slots=torch.from_numpy(slots).to(device),
                max_ctx=int(max_ctx), layout=layout, BS=BS)


def _poison(pr, cache):
    """NaN in the rows the step is going to write."""
    s = pr['slots'][pr['slots'] >= 0]
    blk, row = s // pr['BS'], s % pr['BS']
    for c in (cache[0], cache[1]):
        if pr['layout'] == 'NHSD':
            c[blk, :, row, :] = float('nan')
        else:
            c[blk, row, :, :] = float('nan')


def _both_ways(ops, pr):
    two = pr['cache'].clone()
    _poison(pr, two)
    one = two.clone()
    ops.kv_write_slots(two[0], two[1], pr['kn'], pr['vn'], pr['slots'], layout=pr['layout'])
    want = ops.paged_decode(pr['q'], two[0], two[1], pr['bt'], pr['ctx'], pr['max_ctx'], layout=pr['layout'])
    got = ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], one[0], one[1], pr['bt'], pr['ctx'], pr['max_ctx'],
                                  layout=pr['layout'])
    torch.cuda.synchronize()
    return want, got, two, one


def _assert_same(want, got, two, one):
    assert not torch.isnan(got.float()).any()
    assert torch.equal(want.view(torch.uint8), got.view(torch.uint8))
    assert torch.equal(two.view(torch.uint8), one.view(torch.uint8))   # (bytes: the untouched NaN rows compare equal)


@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('shape', [
    # B, Hq, Hkv, D, BS, max_ctx
    (9, 8, 8, 64, 16, 700),       # one kv head per workgroup, eight sub-ranges
    (5, 8, 4, 64, 16, 300),       # two query heads per kv head on the VALU kernel
    (4, 8, 2, 32, 8, 130), (3, 8, 1, 128, 16, 257), (3, 4, 4, 16, 4, 50), (2, 6, 6, 256, 16, 40),
    (7, 8, 8, 64, 3, 100),        # block size that is not a power of two
    (3, 12, 4, 64, 16, 90),       # three query heads per kv head: two chunks of workgroups write the same rows
    (2, 8, 8, 64, 16, 9000),      # token splits merged in the launch
    (1, 8, 8, 64, 16, 40000),     # more splits than the launch merges itself: merge kernel behind it
    (3, 32, 8, 128, 16, 300),     # grouped-query shape on the matrix cores (16-bit): the write is its own launch
    (40, 8, 8, 64, 16, 33),       # short contexts: the written row is read by the first loads
])
def test_append_equals_write_then_decode(ops, device, shape, dtype, layout):
    B, Hq, Hkv, D, BS, max_ctx = shape
    rng = np.random.default_rng(abs(hash((shape, str(dtype), layout))) % (2 ** 31))
    pr = _problem(rng, device, dtype, B, Hq, Hkv, D, BS, max_ctx, layout)
    _assert_same(*_both_ways(ops, pr))


def test_append_every_short_context_and_block_edge(ops, device):
    """Contexts 0 .. 200 in one batch (empty rows write nothing and give zeros): the written row in
    the first, second and later pipeline steps of its wave, on and off block boundaries."""
    rng = np.random.default_rng(7)
    ctx = np.arange(0, 201)
    for dtype, D, Hkv in ((torch.bfloat16, 64, 8), (torch.float32, 64, 8), (torch.bfloat16, 128, 2)):
        pr = _problem(rng, device, dtype, len(ctx), 8, Hkv, D, 16, 200, 'NHSD', ctx=ctx)
        want, got, two, one = _both_ways(ops, pr)
        _assert_same(want, got, two, one)
        assert float(got[0].float().abs().max()) == 0.0


def test_append_at_the_metric_shape_subset(ops, device):
    """The benchmark's plan (8 heads, head_dim 64, block 16, context 2048, bf16), 64 sequences."""
    rng = np.random.default_rng(11)
    pr = _problem(rng, device, torch.bfloat16, 64, 8, 8, 64, 16, 2048, 'NHSD', ctx=np.full(64, 2048))
    _assert_same(*_both_ways(ops, pr))
    pr = _problem(rng, device, torch.bfloat16, 64, 8, 8, 64, 16, 1024, 'NHSD', ctx=np.full(64, 1024))
    _assert_same(*_both_ways(ops, pr))


def test_append_context_beyond_the_bound_writes_nothing(ops, device):
    rng = np.random.default_rng(3)
    pr = _problem(rng, device, torch.bfloat16, 4, 8, 8, 64, 16, 128, 'NHSD', ctx=[128, 60, 128, 17])
    pr['ctx'][1] = 500                     # beyond max_context: attends to the bound, writes nothing
    before = pr['cache'].clone()
    slots = pr['slots'].clone()
    slots[1] = -1
    pr['slots'] = slots
    want, got, two, one = _both_ways(ops, pr)
    assert torch.equal(want.view(torch.uint8), got.view(torch.uint8))
    assert torch.equal(two.view(torch.uint8), one.view(torch.uint8))
    del before


def test_append_bad_arguments(ops, device):
    rng = np.random.default_rng(5)
    pr = _problem(rng, device, torch.bfloat16, 2, 8, 8, 64, 16, 64, 'NHSD')
    c = pr['cache']
    with pytest.raises(ValueError):
        ops.paged_decode_append(pr['q'], pr['kn'][:, :4], pr['vn'][:, :4], c[0], c[1], pr['bt'], pr['ctx'], 64)
    with pytest.raises(TypeError):
        ops.paged_decode_append(pr['q'], pr['kn'].float(), pr['vn'].float(), c[0], c[1], pr['bt'], pr['ctx'], 64)


def test_append_random_shapes(ops, device):
    """Forty shapes drawn at random (batch, head grouping, head_dim, block size 1-64, ragged contexts
    with empty rows, three dtypes, both layouts): the fused call against the two it replaces."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        Hkv = int(rng.choice([1, 2, 4, 8]))
        gq = int(rng.choice([1, 1, 2, 4, 8]))
        D = int(rng.choice([16, 32, 64, 128, 256]))
        BS = int(rng.choice([1, 2, 4, 8, 16, 16, 32, 64]))
        B = int(rng.integers(1, 40))
        max_ctx = int(rng.integers(1, 700))
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        layout = ['NHSD', 'NSHD'][int(rng.integers(0, 2))]
        ctx = rng.integers(0, max_ctx + 1, size=B)
        ctx[int(rng.integers(0, B))] = max_ctx
        pr = _problem(rng, device, dtype, B, Hkv * gq, Hkv, D, BS, max_ctx, layout, ctx=ctx)
        want, got, two, one = _both_ways(ops, pr)
        assert torch.equal(want.view(torch.uint8), got.view(torch.uint8)), (case, B, Hkv, gq, D, BS, max_ctx, dtype, layout)
        assert torch.equal(two.view(torch.uint8), one.view(torch.uint8)), (case, B, Hkv, gq, D, BS, max_ctx, dtype, layout)
        assert not torch.isnan(got.float()).any()


def test_append_replays_in_a_graph(ops, device):
    """The fused call captured in a HIP graph and replayed on new inputs (same addresses): every
    replay writes the row it is given and attends over it."""
    rng = np.random.default_rng(31)
    for B, ctx_len in ((16, 300), (2, 9000)):          # without and with token splits merged in the launch
        pr = _problem(rng, device, torch.bfloat16, B, 8, 8, 64, 16, ctx_len, 'NHSD', ctx=np.full(B, ctx_len))
        cache = pr['cache'].clone()
        out = torch.empty_like(pr['q'])
        ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], cache[0], cache[1], pr['bt'], pr['ctx'], pr['max_ctx'], out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], cache[0], cache[1], pr['bt'], pr['ctx'], pr['max_ctx'], out=out)
        for _ in range(3):
            pr['kn'].copy_(torch.randn_like(pr['kn']))
            pr['vn'].copy_(torch.randn_like(pr['vn']))
            pr['q'].copy_(torch.randn_like(pr['q']))
            _poison(pr, cache)
            g.replay()
            torch.cuda.synchronize()
            ref = cache.clone()
            _poison(pr, ref)
            ops.kv_write_slots(ref[0], ref[1], pr['kn'], pr['vn'], pr['slots'])
            want = ops.paged_decode(pr['q'], ref[0], ref[1], pr['bt'], pr['ctx'], pr['max_ctx'])
            torch.cuda.synchronize()
            assert torch.equal(want.view(torch.uint8), out.view(torch.uint8))
            assert torch.equal(ref.view(torch.uint8), cache.view(torch.uint8))


# ---------------------------------------------------------------------------
# with the rotary embedding of the step inside the launch (mi355_paged_decode_append_rope)
# ---------------------------------------------------------------------------

def _rope_tables(D, n, dtype, device):
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.outer(torch.arange(n).float(), inv)
    tab = torch.cat([ang, ang], -1)
    return tab.cos().to(dtype).to(device).contiguous(), tab.sin().to(dtype).to(device).contiguous()


def _rope_both_ways(ops, pr, table_len):
    """(rotate q and k in place + cache write) then paged decode, against the one fused launch."""
    D = pr['q'].shape[-1]
    cos_t, sin_t = _rope_tables(D, table_len, pr['q'].dtype, pr['q'].device)
    two = pr['cache'].clone()
    _poison(pr, two)
    one = two.clone()
    q2, k2 = pr['q'].clone(), pr['kn'].clone()
    positions = (pr['ctx'].to(torch.int64) - 1).clamp(min=0)
    ops.rope_kv_write(q2, k2, pr['vn'], cos_t, sin_t, positions, pr['slots'], two[0], two[1], layout=pr['layout'])
    want = ops.paged_decode(q2, two[0], two[1], pr['bt'], pr['ctx'], pr['max_ctx'], layout=pr['layout'])
    q_before = pr['q'].clone()
    got = ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], one[0], one[1], pr['bt'], pr['ctx'], pr['max_ctx'],
                                  layout=pr['layout'], rope=(cos_t, sin_t))
    torch.cuda.synchronize()
    assert torch.equal(q_before, pr['q'])          # the fused call leaves q as it was given
    return want, got, two, one


@pytest.mark.parametrize('layout', ['NHSD', 'NSHD'])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('shape', [
    (9, 8, 8, 64, 16, 700), (5, 8, 4, 64, 16, 300), (4, 8, 2, 32, 8, 130), (3, 8, 1, 128, 16, 257),
    (3, 4, 4, 16, 4, 50), (2, 6, 6, 256, 16, 40), (7, 8, 8, 64, 3, 100), (3, 12, 4, 64, 16, 90),
    (2, 8, 8, 64, 16, 9000), (40, 8, 8, 64, 16, 33), (256, 8, 8, 64, 16, 540),
])
def test_append_with_rope_equals_rope_write_then_decode(ops, device, shape, dtype, layout):
    """Bit for bit, output and caches, over the plans of the VALU kernel (sub-ranges, several heads per
    workgroup, query-head chunks, token splits merged in the launch)."""
    B, Hq, Hkv, D, BS, max_ctx = shape
    if dtype != torch.float32 and D in (64, 128) and Hq // Hkv >= 4:
        pytest.skip('grouped-query shape of the matrix-core kernel')
    rng = np.random.default_rng(abs(hash((shape, str(dtype), layout, 'rope'))) % (2 ** 31))
    pr = _problem(rng, device, dtype, B, Hq, Hkv, D, BS, max_ctx, layout)
    _assert_same(*_rope_both_ways(ops, pr, table_len=max(max_ctx, 8)))


def test_append_with_rope_every_short_context(ops, device):
    rng = np.random.default_rng(17)
    ctx = np.arange(0, 201)
    for dtype, D, Hkv in ((torch.bfloat16, 64, 8), (torch.float32, 32, 4)):
        pr = _problem(rng, device, dtype, len(ctx), 8, Hkv, D, 16, 200, 'NSHD', ctx=ctx)
        want, got, two, one = _rope_both_ways(ops, pr, table_len=256)
        _assert_same(want, got, two, one)
        assert float(got[0].float().abs().max()) == 0.0


def test_append_with_rope_against_the_oracle(ops, device):
    """The fused launch held directly to the CPU oracle (oracle/paged_oracle.py, pinned by the golden
    vectors): rotate-half RoPE of q and the new K row in float64 on the rounded inputs, the rotated
    row rounded to the cache's dtype and placed at position context - 1, then the oracle's paged
    decode.  fp32 at atol = rtol = 1e-5 (the reference's own bar), bf16 at atol 1e-2 / rtol 1.6e-2."""
    from oracle import paged_oracle
    rng = np.random.default_rng(23)
    for dtype, atol, rtol in ((torch.float32, 1e-5, 1e-5), (torch.bfloat16, 1e-2, 1.6e-2)):
        B, H, D, BS, max_ctx = 6, 8, 64, 16, 300
        pr = _problem(rng, device, dtype, B, H, H, D, BS, max_ctx, 'NHSD', ctx=[1, 16, 17, 130, 299, 300])
        cos_t, sin_t = _rope_tables(D, 512, dtype, device)
        cache = pr['cache'].clone()
        _poison(pr, cache)
        got = ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], cache[0], cache[1], pr['bt'], pr['ctx'],
                                      max_ctx, layout='NHSD', rope=(cos_t, sin_t))
        torch.cuda.synchronize()
        ref_cache = pr['cache'].clone().cpu()
        ctx = pr['ctx'].cpu().numpy()
        bt = pr['bt'].cpu().numpy()
        cos64, sin64 = cos_t.cpu().double(), sin_t.cpu().double()

        def rot(x, pos):   # x [H, D] in float64
            lo, hi = x[:, :D // 2], x[:, D // 2:]
            return x * cos64[pos] + torch.cat((-hi, lo), -1) * sin64[pos]

        for b in range(B):
            pos = int(ctx[b]) - 1
            q_rot = rot(pr['q'][b].cpu().double(), pos).to(dtype)
            k_rot = rot(pr['kn'][b].cpu().double(), pos).to(dtype)
            blk, row = int(bt[b, pos // BS]), pos % BS
            ref_cache[0, blk, :, row, :] = k_rot
            ref_cache[1, blk, :, row, :] = pr['vn'][b].cpu()
            want = paged_oracle.paged_decode(q_rot.float().numpy(), ref_cache[0].float().numpy(),
                                             ref_cache[1].float().numpy(), bt[b].tolist(), int(ctx[b]), BS)
            np.testing.assert_allclose(got[b].float().cpu().numpy(), want, atol=atol, rtol=rtol)
            # the cache holds the rotated row (one rounding of the float64 value, up to the last bit of fp32 products)
            np.testing.assert_allclose(cache[0, blk, :, row, :].float().cpu().numpy(), k_rot.float().numpy(),
                                       atol=atol, rtol=rtol)
            assert torch.equal(cache[1, blk, :, row, :].cpu(), pr['vn'][b].cpu())


def test_append_with_rope_refuses_matrix_core_shapes(ops, device):
    from mi355_paged._lib import Mi355Error
    rng = np.random.default_rng(29)
    pr = _problem(rng, device, torch.bfloat16, 3, 32, 8, 128, 16, 300, 'NHSD')
    cos_t, sin_t = _rope_tables(128, 512, torch.bfloat16, device)
    assert not ops.paged_decode_append_rope_serves(torch.bfloat16, 32, 8, 128)
    before = pr['cache'].clone()
    with pytest.raises(Mi355Error):
        ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], pr['cache'][0], pr['cache'][1], pr['bt'], pr['ctx'], 300,
                                rope=(cos_t, sin_t))
    torch.cuda.synchronize()
    assert torch.equal(before.view(torch.uint8), pr['cache'].view(torch.uint8))     # nothing was launched


def test_append_against_the_oracle(ops, device):
    """mi355_paged_decode_append held directly to the CPU oracle (not only to the two calls it replaces):
    the oracle's cache write at position context - 1 followed by its paged decode, both layouts."""
    from oracle import paged_oracle
    rng = np.random.default_rng(31)
    for dtype, layout, atol, rtol in ((torch.float32, 'NHSD', 1e-5, 1e-5), (torch.bfloat16, 'NSHD', 1e-2, 1.6e-2)):
        B, Hq, Hkv, D, BS, max_ctx = 5, 8, 4, 64, 16, 260
        pr = _problem(rng, device, dtype, B, Hq, Hkv, D, BS, max_ctx, layout, ctx=[1, 15, 16, 200, 260])
        cache = pr['cache'].clone()
        _poison(pr, cache)
        got = ops.paged_decode_append(pr['q'], pr['kn'], pr['vn'], cache[0], cache[1], pr['bt'], pr['ctx'], max_ctx,
                                      layout=layout)
        torch.cuda.synchronize()
        ref = pr['cache'].float().cpu()
        if layout == 'NSHD':
            ref = ref.permute(0, 1, 3, 2, 4).contiguous()          # the oracle's layout: [blocks, heads, rows, d]
        kc, vc = ref[0].numpy(), ref[1].numpy()
        paged_oracle.kv_write_slots(kc, vc, pr['kn'].float().cpu().numpy(), pr['vn'].float().cpu().numpy(),
                                    pr['slots'].cpu().tolist(), BS)
        want = paged_oracle.paged_decode_batch(pr['q'].float().cpu().numpy(), kc, vc, pr['bt'].cpu().numpy(),
                                               pr['ctx'].cpu().tolist(), BS)
        np.testing.assert_allclose(got.float().cpu().numpy(), want, atol=atol, rtol=rtol)
        after = cache.float().cpu()
        if layout == 'NSHD':
            after = after.permute(0, 1, 3, 2, 4)
        assert np.array_equal(after[0].numpy(), kc) and np.array_equal(after[1].numpy(), vc)
