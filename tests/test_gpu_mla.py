"""MLA latent-space attention kernel and the drop-in `mini_mla` module (needs an MI355X).

Tolerances: the kernel takes 16-bit inputs, rotates cached k_rope in fp32 and rounds it and the
softmax weights back to 16 bits before the matrix products, accumulating in fp32 ->
atol=1e-2, rtol=1.6e-2 against the float64 oracle on the same rounded inputs (bf16), and
atol=2e-3, rtol=4e-3 for fp16; the fp32 kernel (exact fp32 products on the fp32 matrix
instruction) at atol=2e-5, rtol=1e-4 against the same oracle.  Module outputs with fp32 weights are
compared with the reference module's stored outputs at atol=1e-4 (the attention core of an fp32
module runs the fp32 kernel; the rest is fp32 GEMMs on a different BLAS)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_npz
from oracle import mla_oracle as mo

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: dict(atol=1e-2, rtol=1.6e-2), torch.float16: dict(atol=2e-3, rtol=4e-3)}


@pytest.fixture(scope='module')
def ops():
    from mi355_paged import ops as _ops
    return _ops


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # seqs, H, R, P, max_ctx
    (3, 4, 32, 8, 40), (2, 8, 64, 16, 100), (2, 16, 128, 32, 333), (2, 16, 512, 64, 700),
    (1, 20, 256, 64, 50), (1, 128, 512, 64, 130), (5, 3, 32, 24, 17), (3, 16, 128, 24, 450), (2, 9, 256, 8, 300),
    # more than 16 heads at kv_lora_rank >= 128: 64 heads per workgroup, 16 per wave
    (2, 32, 512, 64, 700), (2, 48, 128, 32, 333), (1, 100, 256, 24, 300), (3, 17, 512, 64, 95), (1, 64, 512, 64, 3000),
])
def test_mla_decode_kernel(ops, device, shape, dtype):
    seqs, H, R, P, max_ctx = shape
    rng = np.random.default_rng(sum(shape))
    S = max_ctx
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(dtype)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[0] = max_ctx
    cos, sin = mo.rope_tables(P, S + 3)
    scale = 0.11
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx, scale)
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), max_ctx, scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[dtype])


@pytest.mark.parametrize('shape', [
    # seqs, H, R, P, max_ctx -- one-wave kernel (rank <= 128) and four-wave kernel (rank >= 256)
    (3, 4, 32, 8, 40), (2, 8, 64, 16, 100), (2, 16, 128, 32, 333), (2, 20, 128, 64, 77), (1, 3, 32, 64, 17),
    (2, 16, 512, 64, 700), (1, 20, 256, 64, 50), (3, 17, 512, 32, 95), (2, 8, 256, 32, 1), (1, 40, 512, 64, 260),
    # few rows over long contexts: token splits + the merge kernel
    (2, 16, 512, 64, 3000), (1, 8, 64, 16, 2000), (3, 4, 128, 32, 1500),
])
def test_mla_decode_fp32_kernel(ops, device, shape):
    seqs, H, R, P, max_ctx = shape
    assert ops.mla_fp32_supported(R, P)
    rng = np.random.default_rng(sum(shape) + 1)
    S = max_ctx
    q = rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5
    c_kv = rng.standard_normal((seqs, S, R), dtype=np.float32)
    k_rope = rng.standard_normal((seqs, S, P), dtype=np.float32)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[0] = max_ctx
    if seqs > 2:
        ctx[1] = 0                                       # an empty context gives a zero row
    cos, sin = mo.rope_tables(P, S + 3)
    scale = 0.11
    live = ctx > 0
    want = np.zeros((seqs, H, R), dtype=np.float64)
    want[live] = mo.latent_attention(q[live], c_kv[live], k_rope[live], cos, sin, ctx[live], scale)
    t = lambda a: torch.from_numpy(a).to(device)
    out = ops.mla_decode(t(q), t(c_kv), t(k_rope), t(cos), t(sin), t(ctx), max_ctx, scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=2e-5, rtol=1e-4)
    # several query tokens of one sequence as rows with their own causal context (row_seq)
    rows = np.array([0, 0, seqs - 1], dtype=np.int32)
    rctx = np.array([max(1, max_ctx // 2), max_ctx, int(max(ctx[seqs - 1], 1))], dtype=np.int32)
    qq = np.ascontiguousarray(np.concatenate([q] * 3)[:3])
    want = mo.latent_attention(qq, c_kv, k_rope, cos, sin, rctx, scale, row_seq=rows)
    out = ops.mla_decode(t(qq), t(c_kv), t(k_rope), t(cos), t(sin), t(rctx), max_ctx, scale, row_seq=t(rows))
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize('shape', [(2, 16, 512, 64, 300), (2, 8, 256, 32, 130), (3, 4, 128, 16, 77), (2, 5, 32, 8, 50)])
def test_mla_decode_fp32_unaligned_rope_rows(ops, device, shape):
    """k_rope rows that do not start on 16-byte boundaries (a view one element into a wider buffer)
    take the fp32 kernel's element-by-element loads; same arithmetic as the 16-byte form, so the
    results are equal bit for bit, and both meet the oracle."""
    seqs, H, R, P, max_ctx = shape
    rng = np.random.default_rng(sum(shape) + 7)
    S = max_ctx
    q = rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5
    c_kv = rng.standard_normal((seqs, S, R), dtype=np.float32)
    k_rope = rng.standard_normal((seqs, S, P), dtype=np.float32)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[0] = max_ctx
    cos, sin = mo.rope_tables(P, S + 3)
    scale = 0.09
    want = mo.latent_attention(q, c_kv, k_rope, cos, sin, ctx, scale)
    t = lambda a: torch.from_numpy(a).to(device)
    aligned = ops.mla_decode(t(q), t(c_kv), t(k_rope), t(cos), t(sin), t(ctx), max_ctx, scale)
    wide = torch.zeros((seqs, S, P + 3), dtype=torch.float32, device=device)
    view = wide[:, :, 1:1 + P]
    view.copy_(t(k_rope))
    assert view.data_ptr() % 16 != 0 and view.stride(2) == 1
    shifted = ops.mla_decode(t(q), t(c_kv), view, t(cos), t(sin), t(ctx), max_ctx, scale)
    torch.cuda.synchronize()
    np.testing.assert_allclose(aligned.cpu().numpy(), want, atol=2e-5, rtol=1e-4)
    assert torch.equal(aligned, shifted)


def test_mla_decode_config4_shape(ops, device):
    """BASELINE config 4: kv_lora_rank=512, rope 64, context 4096 (split path), checked on a few rows."""
    rng = np.random.default_rng(4)
    seqs, H, R, P, S = 4, 16, 512, 64, 4096
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.3).to(torch.bfloat16)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(torch.bfloat16)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(torch.bfloat16)
    ctx = np.array([4096, 4095, 2049, 16], dtype=np.int32)
    cos, sin = mo.rope_tables(P, S)
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx,
                               (128 + 64) ** -0.5)
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, (128 + 64) ** -0.5)
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.bfloat16])


def test_mla_many_heads_kernel_matches_16_head_kernel(ops, device):
    """128 heads over a split context: the kernel that gives each wave its own 16 heads and the one
    that runs a workgroup per 16 heads (knob MI355_MLA_MANY_HEADS = 0) agree within tolerance, and both
    with the oracle on a sample of heads."""
    rng = np.random.default_rng(12)
    seqs, H, R, P, S = 3, 128, 512, 64, 2500
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.3).to(torch.bfloat16)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(torch.bfloat16)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(torch.bfloat16)
    ctx = np.array([S, 1, 1777], dtype=np.int32)
    cos, sin = mo.rope_tables(P, S)
    args = (q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
            torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, 0.07)
    many = ops.mla_decode(*args).cpu().numpy()
    with ops.debug_knob('MI355_MLA_MANY_HEADS', 0):
        tiled = ops.mla_decode(*args).cpu().numpy()
    np.testing.assert_allclose(many, tiled, **TOL[torch.bfloat16])
    heads = [0, 15, 16, 63, 64, 127]
    want = mo.latent_attention(q[:, heads].float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin,
                               ctx, 0.07)
    np.testing.assert_allclose(many[:, heads], want, **TOL[torch.bfloat16])


@pytest.mark.parametrize('H', [16, 48])
def test_mla_scores_that_keep_growing(ops, device, H):
    """Scores that rise along the context by far more than 2^8 in total, with jumps: the many-heads
    kernel moves its softmax reference several times (it defers the rescale of its accumulators
    until a maximum outgrows the reference by 2^8), the 16-head kernel rescales on every rise."""
    rng = np.random.default_rng(H)
    seqs, R, P, S = 2, 512, 64, 1500
    q = rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.1
    c_kv = rng.standard_normal((seqs, S, R), dtype=np.float32) * 0.5
    k_rope = rng.standard_normal((seqs, S, P), dtype=np.float32) * 0.1
    q[:, :, 0] = 2.0
    ramp = np.arange(S, dtype=np.float32) * 0.02
    ramp[700:] += 6.0
    ramp[1203] += 5.0
    c_kv[:, :, 0] = ramp[None, :]
    q, c_kv, k_rope = (torch.from_numpy(x).to(torch.bfloat16) for x in (q, c_kv, k_rope))
    ctx = np.array([S, 1111], dtype=np.int32)
    cos, sin = mo.rope_tables(P, S)
    scale = 1.5      # log2 units: 1.5 * 1.44 * 2 * (0.02 * 1500 + 11) ~ 180 over the context
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx, scale)
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, scale)
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.bfloat16])


def test_mla_decode_large_shape_properties(ops, device):
    """Config-4 geometry at serving size (64 rows x 4096 tokens, kv_lora_rank 512, rope 64, bf16):
    rows against the float64 oracle, row order does not matter (bit for bit), and a row read
    through row_seq from another position of the cache tensor gives the same bits."""
    rows, H, R, P, S = 64, 16, 512, 64, 4096
    g = torch.Generator(device=device).manual_seed(8)
    q = (torch.randn((rows, H, R + P), generator=g, device=device) * 0.3).to(torch.bfloat16)
    c_kv = torch.randn((rows, S, R), generator=g, device=device).to(torch.bfloat16)
    k_rope = torch.randn((rows, S, P), generator=g, device=device).to(torch.bfloat16)
    ctx = torch.randint(S // 2, S + 1, (rows,), generator=g, device=device, dtype=torch.int32)
    ctx[5] = S
    cos, sin = mo.rope_tables(P, S)
    cosd, sind = torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device)
    scale = (128 + 64) ** -0.5
    out = ops.mla_decode(q, c_kv, k_rope, cosd, sind, ctx, S, scale)
    pick = [0, 5, 63]
    want = mo.latent_attention(q[pick].float().cpu().numpy(), c_kv[pick].float().cpu().numpy(),
                               k_rope[pick].float().cpu().numpy(), cos, sin, ctx[pick].cpu().numpy(), scale)
    np.testing.assert_allclose(out[pick].cpu().numpy(), want, **TOL[torch.bfloat16])
    perm = torch.randperm(rows, generator=g, device=device)
    out_perm = ops.mla_decode(q[perm], c_kv, k_rope, cosd, sind, ctx[perm], S, scale, row_seq=perm.to(torch.int32))
    assert torch.equal(out_perm, out[perm])


def test_mla_causal_rows_of_one_sequence(ops, device):
    """T query tokens of one sequence as T rows (row_seq) with growing contexts = causal prefill."""
    rng = np.random.default_rng(9)
    H, R, P, S, T = 8, 64, 16, 60, 23
    q = torch.from_numpy(rng.standard_normal((2 * T, H, R + P), dtype=np.float32) * 0.5).to(torch.float16)
    c_kv = torch.from_numpy(rng.standard_normal((2, S, R), dtype=np.float32)).to(torch.float16)
    k_rope = torch.from_numpy(rng.standard_normal((2, S, P), dtype=np.float32)).to(torch.float16)
    ctx = np.concatenate([np.arange(S - T, S) + 1, np.arange(T) + 1]).astype(np.int32)
    row_seq = np.repeat([0, 1], T).astype(np.int32)
    cos, sin = mo.rope_tables(P, S)
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx, 0.2,
                               row_seq)
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, 0.2,
                         row_seq=torch.from_numpy(row_seq).to(device))
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.float16])


# ------------------------------------------------------------------ drop-in mini_mla module
@pytest.fixture(scope='module')
def mla_mod(device):
    import mini_mla
    return mini_mla


def test_module_matches_reference_outputs(mla_mod, device):
    """Golden G9: same weights and inputs as the reference module, on host and device tensors."""
    z = load_npz('g9_mla.npz')
    meta = json.loads(str(z['meta']))
    for ci, m in enumerate(meta['models']):
        for dev in ('cpu', 'cuda'):
            mod = mla_mod.MultiHeadLatentAttention(**m['config']).eval()
            state = {k[len(f'm{ci}_w::'):]: torch.from_numpy(z[k].astype(np.float32))
                     for k in z.files if k.startswith(f'm{ci}_w::')}
            missing, unexpected = mod.load_state_dict(state, strict=False)
            assert not unexpected and all('rotary' in k for k in missing)
            mod = mod.to(dev)
            for name in m['calls']:
                pre = f'm{ci}_{name}_'
                t = lambda a: torch.from_numpy(z[a]).to(dev)
                past = (t(pre + 'past_c_kv'), t(pre + 'past_k_rope')) if pre + 'past_c_kv' in z.files else (None, None)
                with torch.no_grad():
                    out, nc, nr = mod(t(pre + 'x'), t(pre + 'pos'), *past)
                np.testing.assert_allclose(out.cpu().numpy(), z[pre + 'out'], atol=1e-4, rtol=1e-4)
                np.testing.assert_allclose(nc.cpu().numpy(), z[pre + 'new_c_kv'], atol=1e-5, rtol=1e-4)
                np.testing.assert_allclose(nr.cpu().numpy(), z[pre + 'new_k_rope'], atol=1e-5, rtol=1e-4)
    assert mla_mod.compare_kv_cache_size(num_heads=32, head_dim=128, kv_lora_rank=512, qk_rope_head_dim=64,
                                         num_layers=32, seq_len=4096, batch_size=1) == tuple(meta['kv_size_mb'])


class TestReferenceSuite:
    """04_mla/test_mini_mla.py:10-170 restated against the drop-in module."""

    def test_rotary(self, mla_mod):
        rope = mla_mod.RotaryEmbedding(dim=64, max_seq_len=512)
        for length in (1, 10, 128, 512, 600):
            cos, sin = rope(length)
            assert cos.shape == (length, 64) and sin.shape == (length, 64)
            assert torch.allclose(cos ** 2 + sin ** 2, torch.ones_like(cos), atol=1e-5)
        cos, sin = rope(1)
        assert torch.allclose(cos[0], torch.ones(64), atol=1e-5) and torch.allclose(sin[0], torch.zeros(64), atol=1e-5)

    def test_rotate_half(self, mla_mod):
        x = torch.randn(3, 16)
        assert mla_mod.rotate_half(torch.randn(4, 8, 64)).shape == (4, 8, 64)
        assert torch.allclose(mla_mod.rotate_half(mla_mod.rotate_half(x)), -x, atol=1e-6)
        q, k = torch.randn(1, 2, 5, 8), torch.randn(1, 2, 5, 8)
        cos, sin = mla_mod.RotaryEmbedding(8)(5)
        qr, kr = mla_mod.apply_rotary_emb(q, k, cos, sin)
        assert torch.allclose(qr.norm(dim=-1), q.norm(dim=-1), atol=1e-5) and kr.shape == k.shape

    @pytest.fixture
    def small_mla(self, mla_mod):
        return mla_mod.MultiHeadLatentAttention(hidden_size=128, num_heads=4, qk_nope_head_dim=16,
                                                qk_rope_head_dim=8, kv_lora_rank=32, v_head_dim=16, max_seq_len=256)

    def test_prefill_and_decode_shapes(self, small_mla):
        B, T, D = 2, 16, 128
        out, c_kv, k_rope = small_mla(torch.randn(B, T, D), torch.arange(T).unsqueeze(0).expand(B, -1))
        assert out.shape == (B, T, D) and c_kv.shape == (B, T, 32) and k_rope.shape == (B, T, 8)
        out, nc, nr = small_mla(torch.randn(B, 1, D), torch.tensor([[T]] * B), past_c_kv=c_kv, past_k_rope=k_rope)
        assert out.shape == (B, 1, D) and nc.shape == (B, 1, 32) and nr.shape == (B, 1, 8)

    def test_kv_cache_accumulation(self, small_mla, mla_mod):
        _, c_kv, k_rope = small_mla(torch.randn(1, 5, 128), torch.arange(5).unsqueeze(0))
        for step in range(3):
            _, nc, nr = small_mla(torch.randn(1, 1, 128), torch.tensor([[5 + step]]), past_c_kv=c_kv,
                                  past_k_rope=k_rope)
            c_kv, k_rope = torch.cat([c_kv, nc], 1), torch.cat([k_rope, nr], 1)
        assert c_kv.shape[1] == 8
        cache = mla_mod.MLAKVCache(2, 16, 1, 32, 8, torch.device('cpu'), torch.float32)
        cache.update(0, c_kv.detach(), k_rope.detach(), None)
        got_c, got_r = cache.get(0, 1)
        assert torch.equal(got_c[0], c_kv[0].detach()) and torch.equal(got_r[0], k_rope[0].detach())

    def test_gradient_backprop(self, small_mla):
        x = torch.randn(2, 8, 128, requires_grad=True)
        out, _, _ = small_mla(x, torch.arange(8).unsqueeze(0).expand(2, -1))
        out.sum().backward()
        assert x.grad is not None and not torch.isnan(x.grad).any() and x.grad.abs().sum() > 0
        assert small_mla.kv_b_proj.weight.grad is not None

    def test_causal_mask_applied_prefill(self, small_mla):
        small_mla.eval()
        x = torch.randn(1, 4, 128)
        pos = torch.arange(4).unsqueeze(0)
        with torch.no_grad():
            out1, _, _ = small_mla(x, pos)
            x2 = x.clone()
            x2[0, 2:] = torch.randn_like(x2[0, 2:])
            out2, _, _ = small_mla(x2, pos)
        assert torch.allclose(out1[0, :2], out2[0, :2], atol=1e-4)

    def test_kv_size_helper(self, mla_mod):
        mha, mla = mla_mod.compare_kv_cache_size(num_heads=32, head_dim=128, kv_lora_rank=512, qk_rope_head_dim=64,
                                                 num_layers=32, seq_len=4096, batch_size=1)
        assert mha / mla > 5
        _, a = mla_mod.compare_kv_cache_size(seq_len=1024, num_heads=16, head_dim=64, kv_lora_rank=128,
                                             qk_rope_head_dim=32, num_layers=8, batch_size=1)
        _, b = mla_mod.compare_kv_cache_size(seq_len=4096, num_heads=16, head_dim=64, kv_lora_rank=128,
                                             qk_rope_head_dim=32, num_layers=8, batch_size=1)
        assert abs(b / a - 4.0) < 0.01


def test_module_bf16_on_device_config4_like(mla_mod, device):
    """A bf16 module with kv_lora_rank=512 / rope 64 decoding against a 1000-token cache: the module's
    output equals the oracle's expanded-form forward on the same (bf16-rounded) weights."""
    torch.manual_seed(3)
    cfg = dict(hidden_size=256, num_heads=16, qk_nope_head_dim=128, qk_rope_head_dim=64, kv_lora_rank=512,
               v_head_dim=128, max_seq_len=2048)
    mod = mla_mod.MultiHeadLatentAttention(**cfg).to(torch.bfloat16).eval().to(device)
    B, S = 2, 1000
    past_c = (torch.randn(B, S, 512) * 0.5).to(torch.bfloat16).to(device)
    past_r = (torch.randn(B, S, 64) * 0.5).to(torch.bfloat16).to(device)
    x = (torch.randn(B, 1, 256) * 0.5).to(torch.bfloat16).to(device)
    pos = torch.full((B, 1), S, device=device)
    with torch.no_grad():
        out, nc, nr = mod(x, pos, past_c, past_r)
    w = {k: v.float().cpu().numpy() for k, v in mod.state_dict().items() if 'rotary' not in k}
    want, _, _ = mo.mla_forward(w, cfg, x.float().cpu().numpy(), pos.cpu().numpy(), past_c.float().cpu().numpy(),
                                past_r.float().cpu().numpy(), dtype=np.float64)
    np.testing.assert_allclose(out.float().cpu().numpy(), want, atol=3e-2, rtol=5e-2)


def test_mla_decode_rejects_contexts_beyond_the_cache(ops, device):
    """The kernel reads max_context cache rows and rotary table rows per sequence: a bound larger
    than either, or per-row vectors of the wrong length, are refused before the launch."""
    q = torch.zeros((2, 4, 32 + 8), dtype=torch.bfloat16, device=device)
    c_kv = torch.zeros((2, 10, 32), dtype=torch.bfloat16, device=device)
    k_rope = torch.zeros((2, 10, 8), dtype=torch.bfloat16, device=device)
    cos, sin = (torch.from_numpy(t).to(device) for t in mo.rope_tables(8, 10))
    ctx = torch.tensor([10, 3], dtype=torch.int32, device=device)
    ops.mla_decode(q, c_kv, k_rope, cos, sin, ctx, 10, 0.1)
    with pytest.raises(ValueError):
        ops.mla_decode(q, c_kv, k_rope, cos, sin, ctx, 11, 0.1)
    with pytest.raises(ValueError):
        ops.mla_decode(q, c_kv, k_rope, cos[:6], sin[:6], ctx, 10, 0.1)
    with pytest.raises(ValueError):
        ops.mla_decode(q, c_kv, k_rope[:, :9], cos, sin, ctx, 9, 0.1)
    with pytest.raises(ValueError):
        ops.mla_decode(q, c_kv, k_rope, cos, sin, ctx[:1], 10, 0.1)
    with pytest.raises(ValueError):
        ops.mla_decode(q, c_kv[:1], k_rope[:1], cos, sin, ctx, 10, 0.1)


# ------------------------------------------------------------------ block cache (mi355_mla_decode_paged)
def _scatter_into_blocks(rng, c_kv, k_rope, block_size, spare=3):
    """A dense cache [seqs, S, .] copied into randomly chosen blocks of one pool [blocks, block_size, R + P]
    whose other rows hold NaN; returns (pool, block_tables)."""
    seqs, S, R = c_kv.shape
    P = k_rope.shape[2]
    per = (S + block_size - 1) // block_size
    NB = seqs * per + spare
    bt = rng.permutation(NB)[:seqs * per].reshape(seqs, per).astype(np.int32)
    pool = torch.full((NB, block_size, R + P), float('nan'), dtype=c_kv.dtype)
    for s in range(seqs):
        for j in range(per):
            n = min(block_size, S - j * block_size)
            pool[bt[s, j], :n, :R] = c_kv[s, j * block_size:j * block_size + n]
            pool[bt[s, j], :n, R:] = k_rope[s, j * block_size:j * block_size + n]
    return pool, bt


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [
    # seqs, H, R, P, max_ctx, block_size
    (3, 4, 32, 8, 40, 4), (2, 8, 64, 16, 100, 16), (2, 16, 128, 32, 333, 16), (2, 16, 512, 64, 700, 16),
    (1, 20, 256, 64, 50, 8), (3, 17, 512, 64, 95, 32), (2, 32, 512, 64, 700, 16), (1, 64, 512, 64, 3000, 16),
    (2, 16, 512, 64, 130, 1), (1, 16, 512, 64, 9000, 64), (3, 16, 512, 64, 333, 8), (2, 48, 512, 64, 1500, 128),
    (2, 16, 512, 64, 97, 4), (2, 16, 512, 32, 200, 16),
])
def test_mla_decode_paged_equals_the_dense_call(ops, device, shape, dtype):
    """A dense latent cache scattered into random blocks of a pool (every other pool row NaN): the
    block-table call gives the dense call's result bit for bit, for every kernel (one wave, four
    waves, 64 heads per workgroup, fp32; token splits), and both equal the oracle's gather form."""
    seqs, H, R, P, max_ctx, bs = shape
    if dtype == torch.float32 and not ops.mla_fp32_supported(R, P):
        pytest.skip('rank / rope dim outside the fp32 kernel')
    rng = np.random.default_rng(sum(shape) + 1)
    S = max_ctx
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(dtype)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[0] = max_ctx
    cos, sin = mo.rope_tables(P, S + 3)
    pool, bt = _scatter_into_blocks(rng, c_kv, k_rope, bs)
    cos_d, sin_d, ctx_d = torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device)
    dense = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), cos_d, sin_d, ctx_d, max_ctx, 0.11)
    pool_d = pool.to(device)
    paged = ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bt).to(device),
                                 cos_d, sin_d, ctx_d, max_ctx, 0.11)
    torch.cuda.synchronize()
    assert not torch.isnan(paged).any()
    assert torch.equal(paged.view(torch.int32), dense.view(torch.int32))
    want = mo.latent_attention_paged(q.float().numpy(), pool[..., :R].float().numpy(), pool[..., R:].float().numpy(),
                                     bt, bs, cos, sin, ctx, 0.11)
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else TOL[dtype]
    np.testing.assert_allclose(paged.cpu().numpy(), want, **tol)


def test_mla_decode_paged_shared_tables_bad_ids_and_arguments(ops, device):
    """Several query rows on one table row (row_seq: the causal rows of a chunk), block ids outside the
    pool (clamped: no fault, other rows unaffected), zero-length rows, and refused arguments."""
    from mi355_paged._lib import Mi355Error
    rng = np.random.default_rng(99)
    H, R, P, S, T, bs = 8, 128, 32, 90, 7, 16
    q = torch.from_numpy(rng.standard_normal((2 * T, H, R + P), dtype=np.float32) * 0.5).to(torch.bfloat16)
    c_kv = torch.from_numpy(rng.standard_normal((2, S, R), dtype=np.float32)).to(torch.bfloat16)
    k_rope = torch.from_numpy(rng.standard_normal((2, S, P), dtype=np.float32)).to(torch.bfloat16)
    ctx = np.concatenate([np.arange(S - T, S) + 1, np.arange(T)]).astype(np.int32)       # (second chunk: contexts 0 .. T-1)
    row_seq = np.repeat([0, 1], T).astype(np.int32)
    cos, sin = mo.rope_tables(P, S)
    pool, bt = _scatter_into_blocks(rng, c_kv, k_rope, bs)
    want = mo.latent_attention_paged(q.float().numpy(), pool[..., :R].float().numpy(), pool[..., R:].float().numpy(),
                                     bt, bs, cos, sin, ctx, 0.2, row_seq)
    pool_d = torch.nan_to_num(pool, nan=0.0).to(device)
    args = (torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, 0.2)
    out = ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bt).to(device), *args,
                               row_seq=torch.from_numpy(row_seq).to(device))
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.bfloat16])
    assert float(out[T].abs().max()) == 0.0                                              # context 0: zeros
    bad = bt.copy()
    bad[1, :] = [2 ** 30, -5, 10 ** 6, -1, 7, 123456][:bad.shape[1]]
    out_bad = ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bad).to(device), *args,
                                   row_seq=torch.from_numpy(row_seq).to(device))
    torch.cuda.synchronize()
    assert torch.equal(out_bad[:T], out[:T]) and torch.isfinite(out_bad).all()
    with pytest.raises(Mi355Error):      # block size that is not a power of two
        p3 = torch.zeros((4, 12, R + P), dtype=torch.bfloat16, device=device)
        ops.mla_decode_paged(q.to(device), p3[..., :R], p3[..., R:], torch.zeros((2, 8), dtype=torch.int32, device=device),
                             *args, row_seq=torch.from_numpy(row_seq).to(device))
    with pytest.raises(TypeError):
        ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bt).long().to(device), *args)


def test_module_over_a_block_cache_matches_reference_outputs(mla_mod, device):
    """Golden G9 through forward_paged: the rows the reference was handed as dense past_c_kv /
    past_k_rope tensors are put into a block cache first (blocks drawn from a BlockAllocator in
    scrambled order), the call's own rows are appended by forward_paged, and the output must be
    the reference module's (fp32 weights, fp32 pool: atol 1e-4 like the dense module test)."""
    import mini_vllm
    z = load_npz('g9_mla.npz')
    meta = json.loads(str(z['meta']))
    checked = 0
    for ci, m in enumerate(meta['models']):
        cfg = m['config']
        if not ops_fp32_ok(cfg):
            continue
        mod = mla_mod.MultiHeadLatentAttention(**cfg).eval()
        state = {k[len(f'm{ci}_w::'):]: torch.from_numpy(z[k].astype(np.float32))
                 for k in z.files if k.startswith(f'm{ci}_w::')}
        mod.load_state_dict(state, strict=False)
        mod = mod.to(device)
        for name in m['calls']:
            pre = f'm{ci}_{name}_'
            t = lambda a: torch.from_numpy(z[a]).to(device)
            x, pos = t(pre + 'x'), t(pre + 'pos')
            B = x.shape[0]
            alloc = mini_vllm.BlockAllocator(64, 8)
            for _ in range(5):                      # scramble the free list
                ids = [alloc.allocate() for _ in range(7)]
                for i in ids[::2] + ids[1::2]:
                    alloc.free(i)
            cache = mla_mod.PagedMLAKVCache(64, 8, 1, cfg['kv_lora_rank'], cfg['qk_rope_head_dim'], device,
                                            dtype=torch.float32, allocator=alloc)
            if pre + 'past_c_kv' in z.files:
                pc, pk = t(pre + 'past_c_kv'), t(pre + 'past_k_rope')
                for b in range(B):
                    cache.append(('seq', b), 0, pc[b], pk[b])
            out = mod.forward_paged(x, pos, cache, 0, [('seq', b) for b in range(B)])
            np.testing.assert_allclose(out.cpu().numpy(), z[pre + 'out'], atol=1e-4, rtol=1e-4)
            free_before = alloc.num_free
            for b in range(B):
                cache.free(('seq', b))
            assert alloc.num_free > free_before and alloc.num_free == 64
            checked += 1
    assert checked >= 2


def ops_fp32_ok(cfg):
    from mi355_paged import ops as _ops
    return _ops.mla_fp32_supported(cfg['kv_lora_rank'], cfg['qk_rope_head_dim'])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # seqs, H, max_ctx, contexts
    (3, 128, 2500, [2500, 1, 1777]), (2, 65, 130, [130, 33]), (4, 96, 700, [700, 0, 32, 31]),
    (2, 200, 300, [300, 64]), (1, 128, 9000, [9000]), (5, 128, 96, [96, 95, 65, 64, 63]),
    (2, 17, 200, [200, 77]), (3, 40, 1100, [1100, 512, 3]), (2, 64, 2049, [2049, 2048]),
])
def test_wide_kernel_matches_64_head_kernel_and_oracle(ops, device, shape, dtype):
    """More than 16 heads at kv_lora_rank 512 / rope 64: the kernel that runs 128 heads per workgroup
    (32 per wave, tiles loaded straight into LDS) against the 64-heads-per-workgroup kernel (knob
    MI355_MLA_WIDE = 0) within tolerance and against the oracle on a sample of heads: head counts
    that leave waves or lanes without a head, several head tiles, token splits, contexts of 0, 1 and
    around the 32-token tile, and a non-positive scale (which the wide kernel hands to the other)."""
    seqs, H, S, ctxs = shape
    R, P = 512, 64
    rng = np.random.default_rng(H + S)
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.3).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(dtype)
    ctx = np.array(ctxs, dtype=np.int32)
    cos, sin = mo.rope_tables(P, S)
    args = (q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
            torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, 0.07)
    with ops.debug_knob('MI355_MLA_WIDE_SPLIT_TOKENS', 1):      # (the plan leaves short token splits to the 64-head kernel)
        wide = ops.mla_decode(*args).cpu().numpy()
    with ops.debug_knob('MI355_MLA_WIDE', 0):
        ref = ops.mla_decode(*args).cpu().numpy()
    assert not np.isnan(wide).any()
    np.testing.assert_allclose(wide, ref, **TOL[dtype])
    heads = sorted({0, 31 % H, 32 % H, 63 % H, 64 % H, H - 1, H // 2})
    want = mo.latent_attention(q[:, heads].float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin,
                               ctx, 0.07)
    np.testing.assert_allclose(wide[:, heads], want, **TOL[dtype])
    if seqs == 2 and H == 65:      # a scale that is not positive: served by the 64-head kernel
        neg = ops.mla_decode(*args[:-1], -0.07).cpu().numpy()
        want = mo.latent_attention(q[:, heads].float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos,
                                   sin, ctx, -0.07)
        np.testing.assert_allclose(neg[:, heads], want, **TOL[dtype])


def test_wide_kernel_scores_that_keep_growing(ops, device):
    """Scores that rise along the context by far more than 2^8 in total, with jumps: the kernel moves
    its softmax reference (and rescales its 256 accumulator registers) several times."""
    H = 128
    rng = np.random.default_rng(H)
    seqs, R, P, S = 2, 512, 64, 1500
    q = rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.1
    c_kv = rng.standard_normal((seqs, S, R), dtype=np.float32) * 0.5
    k_rope = rng.standard_normal((seqs, S, P), dtype=np.float32) * 0.1
    q[:, :, 0] = 2.0
    ramp = np.arange(S, dtype=np.float32) * 0.02
    ramp[700:] += 6.0
    ramp[1203] += 5.0
    c_kv[:, :, 0] = ramp[None, :]
    q, c_kv, k_rope = (torch.from_numpy(x).to(torch.bfloat16) for x in (q, c_kv, k_rope))
    ctx = np.array([S, 1111], dtype=np.int32)
    cos, sin = mo.rope_tables(P, S)
    scale = 1.5
    want = mo.latent_attention(q.float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin, ctx, scale)
    with ops.debug_knob('MI355_MLA_WIDE_SPLIT_TOKENS', 1):
        out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                             torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, scale)
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.bfloat16])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 128, 700, 16), (1, 100, 3000, 64), (3, 128, 130, 1), (2, 72, 1500, 8), (2, 128, 333, 128),
                                   (1, 128, 2500, 1), (2, 32, 900, 16), (3, 48, 1500, 8), (2, 20, 260, 32)])
def test_wide_kernel_paged_equals_dense(ops, device, shape, dtype):
    """The 128-heads-per-workgroup kernel over a block cache: the dense call's result bit for bit,
    for block sizes 1 .. 128 (ids staged in LDS per workgroup; a token range with more blocks than fit
    there -- blocks of 1 token -- goes to the 64-head kernel and is compared within tolerance)."""
    seqs, H, max_ctx, bs = shape
    R, P = 512, 64
    rng = np.random.default_rng(sum(shape) + 1)
    S = max_ctx
    q = torch.from_numpy(rng.standard_normal((seqs, H, R + P), dtype=np.float32) * 0.5).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((seqs, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((seqs, S, P), dtype=np.float32)).to(dtype)
    ctx = rng.integers(1, max_ctx + 1, size=seqs).astype(np.int32)
    ctx[0] = max_ctx
    cos, sin = mo.rope_tables(P, S + 3)
    pool, bt = _scatter_into_blocks(rng, c_kv, k_rope, bs)
    cos_d, sin_d, ctx_d = torch.from_numpy(cos).to(device), torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device)
    with ops.debug_knob('MI355_MLA_WIDE_SPLIT_TOKENS', 1):
        dense = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), cos_d, sin_d, ctx_d, max_ctx, 0.11)
        pool_d = pool.to(device)
        paged = ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bt).to(device),
                                     cos_d, sin_d, ctx_d, max_ctx, 0.11)
    torch.cuda.synchronize()
    assert not torch.isnan(paged).any()
    if bs * 1982 > max_ctx:
        assert torch.equal(paged.view(torch.int32), dense.view(torch.int32))
    else:
        np.testing.assert_allclose(paged.cpu().numpy(), dense.cpu().numpy(), **TOL[dtype])


def _where_unequal(got, want):
    idx = (got.view(torch.int32) != want.view(torch.int32)).nonzero()
    if not len(idx):
        return 'equal'
    r, hh, d = idx[0].tolist()
    return (f"{len(idx)} values in rows {sorted(set(idx[:, 0].tolist()))[:8]}, heads {sorted(set(idx[:, 1].tolist()))[:16]}, "
            f"dims {int(idx[:, 2].min())}..{int(idx[:, 2].max())}; first ({r}, {hh}, {d}): {float(got[r, hh, d])} vs {float(want[r, hh, d])}")


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 128, 2100, 6, 0), (5, 40, 700, 10, 0), (2, 200, 1300, 8, 0), (4, 128, 4096, 8, 16),
                                   (128, 128, 2300, 0, 0), (4, 24, 1300, 8, 0), (3, 31, 2100, 6, 16), (6, 64, 1100, 12, 0)])
def test_wide_kernel_merges_two_token_splits_in_the_launch(ops, device, shape, dtype):
    """Two token splits of the 128-heads-per-workgroup kernel: the workgroup of a pair that finishes
    later combines its registers with the other's record (MI355_MLA_WIDE_FUSED_MERGE, the default) --
    bit for bit what the merge kernel makes of the two records (knob = 0), call after call on one
    workspace whose contents are scrambled in between (the arrival cells must not depend on what it
    held), rows whose second split is empty, an empty row, over a block cache, and replays of one
    captured launch; and, within tolerance, with the first split shortened (the default: the second
    workgroup then finds the first one's record written and writes none itself)."""
    seqs, H, S, target, bs = shape
    R, P = 512, 64
    g = torch.Generator(device=device).manual_seed(seqs + H + S)
    c_kv = torch.randn((seqs, S, R), generator=g, device=device).to(dtype)
    k_rope = torch.randn((seqs, S, P), generator=g, device=device).to(dtype)
    cos, sin = (torch.from_numpy(x).to(device) for x in mo.rope_tables(P, S))
    if bs:
        per = -(-S // bs)
        order = torch.randperm(seqs * per, generator=g, device=device)
        pool = torch.zeros((seqs * per, bs, R + P), dtype=dtype, device=device)
        rows = torch.nn.functional.pad(torch.cat((c_kv, k_rope), -1), (0, 0, 0, per * bs - S))
        pool[order] = rows.view(seqs * per, bs, R + P)
        bt = order.view(seqs, per).to(torch.int32).contiguous()

    def call(q, ctx, out=None):
        if bs:
            return ops.mla_decode_paged(q, pool[..., :R], pool[..., R:], bt, cos, sin, ctx, S, 0.09, out=out)
        return ops.mla_decode(q, c_kv, k_rope, cos, sin, ctx, S, 0.09, out=out)

    code = ops._DTYPE_CODE[dtype]
    stream = torch.cuda.current_stream(device).cuda_stream
    def even(q, ctx, out=None):     # the two splits where the merge kernel's plan puts them
        with ops.debug_knob('MI355_MLA_WIDE_SKEW_TOKENS', 0):
            return call(q, ctx, out=out)

    with ops.debug_knob('MI355_MLA_WIDE_SPLIT_TOKENS', 1), ops.debug_knob('MI355_MLA_TARGET_WGS', target):
        need = ops._lib.load().mi355_mla_decode_workspace_bytes(seqs, H, R, P, S, code)
        with ops.debug_knob('MI355_MLA_WIDE_FUSED_MERGE', 0):
            plain = ops._lib.load().mi355_mla_decode_workspace_bytes(seqs, H, R, P, S, code)
        assert need == plain + seqs * ((H + 127) // 128) * 8, "the case is meant to run two splits, merged in the launch"
        for it in range(8):
            q = (torch.randn((seqs, H, R + P), generator=g, device=device) * 0.4).to(dtype)
            ctx = torch.randint(1, S + 1, (seqs,), generator=g, device=device).to(torch.int32)
            ctx[it % seqs] = S if it % 3 else 0
            ctx[(it + 1) % seqs] = 1 + it
            with ops.debug_knob('MI355_MLA_WIDE_FUSED_MERGE', 0):
                want = call(q, ctx)
            ws, _ = ops._workspace(device, need, stream)
            if it % 4 == 0:
                ws.zero_()
            elif it % 4 == 1:
                ws.fill_(255)
            elif it % 4 == 2:
                ws.view(torch.int64)[:need // 8].random_(-2 ** 62, 2 ** 62, generator=g)
            got = even(q, ctx)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), f"call {it}: {_where_unequal(got, want)}"
            assert not got[it % seqs].abs().sum().item() if not it % 3 else True
            # split 0 shortened so that only one workgroup of a pair writes a record: other split
            # points, the same attention
            with ops.debug_knob('MI355_MLA_WIDE_SKEW_TOKENS', 64 if it % 2 else 160):
                skewed = call(q, ctx)
            np.testing.assert_allclose(skewed.cpu().numpy(), want.cpu().numpy(), **TOL[dtype])

        q = (torch.randn((seqs, H, R + P), generator=g, device=device) * 0.4).to(dtype)
        ctx = torch.full((seqs,), S, dtype=torch.int32, device=device)
        out = torch.empty((seqs, H, R), dtype=torch.float32, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            even(q, ctx, out=out)
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            even(q, ctx, out=out)
        for it in range(4):
            q.copy_((torch.randn((seqs, H, R + P), generator=g, device=device) * 0.4).to(dtype))
            ctx.copy_(torch.randint(1, S + 1, (seqs,), generator=g, device=device).to(torch.int32))
            graph.replay()
            with ops.debug_knob('MI355_MLA_WIDE_FUSED_MERGE', 0):
                want = call(q, ctx)
            assert torch.equal(out.view(torch.int32), want.view(torch.int32)), f"replay {it}"


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape', [(1, 128, 20000), (3, 16, 30000), (2, 64, 9000), (2, 16, 12000, 128, 64), (1, 32, 9000, 256, 32)])
def test_many_splits_of_few_rows_match_the_oracle(ops, device, shape, dtype):
    """Few rows with long contexts run as hundreds of token splits, combined by the kernel that spreads a
    row's splits over the threads of a workgroup: against the oracle on a sample of heads, with a full
    row, a row that fills only some of the splits, and an empty one."""
    rows, H, S = shape[:3]
    R, P = shape[3:] if len(shape) > 3 else (512, 64)
    if dtype == torch.float32 and not ops.mla_fp32_supported(R, P):
        pytest.skip("no fp32 kernel for this rank")
    rng = np.random.default_rng(rows + H + S)
    q = torch.from_numpy(rng.standard_normal((rows, H, R + P), dtype=np.float32) * 0.3).to(dtype)
    c_kv = torch.from_numpy(rng.standard_normal((1, S, R), dtype=np.float32)).to(dtype)
    k_rope = torch.from_numpy(rng.standard_normal((1, S, P), dtype=np.float32)).to(dtype)
    ctx = np.array(([S, S // 7, 0] * rows)[:rows], dtype=np.int32)
    row_seq = torch.zeros((rows,), dtype=torch.int32, device=device)
    cos, sin = mo.rope_tables(P, S)
    code = ops._DTYPE_CODE[dtype]
    assert ops._lib.load().mi355_mla_decode_workspace_bytes(rows, H, R, P, S, code) >= rows * H * 8 * (R + 2) * 4, \
        "the case is meant to run at least eight splits"
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, 0.06,
                         row_seq=row_seq).cpu().numpy()
    heads = sorted({0, H // 2, H - 1})
    want = mo.latent_attention(q[:, heads].float().numpy(), np.repeat(c_kv.float().numpy(), rows, 0),
                               np.repeat(k_rope.float().numpy(), rows, 0), cos, sin, ctx, 0.06)
    tol = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else TOL[dtype]
    np.testing.assert_allclose(out[:, heads], want, **tol)
    assert not out[ctx == 0].any()
