"""Work-in-progress checks of the 128-heads-per-workgroup MLA kernel (run by naming the file)."""
import numpy as np
import pytest
import torch

from oracle import mla_oracle as mo
from test_gpu_mla import TOL, _scatter_into_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from mi355_paged import ops as _ops
    return _ops


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [
    # seqs, H, max_ctx, contexts
    (3, 128, 2500, [2500, 1, 1777]), (2, 65, 130, [130, 33]), (4, 96, 700, [700, 0, 32, 31]),
    (2, 200, 300, [300, 64]), (1, 128, 9000, [9000]), (5, 128, 96, [96, 95, 65, 64, 63]),
])
def test_wide_kernel_matches_64_head_kernel_and_oracle(ops, device, shape, dtype):
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
    wide = ops.mla_decode(*args).cpu().numpy()
    with ops.debug_knob('MI355_MLA_WIDE', 0):
        ref = ops.mla_decode(*args).cpu().numpy()
    assert not np.isnan(wide).any()
    np.testing.assert_allclose(wide, ref, **TOL[dtype])
    heads = sorted({0, 31, 32, 63, 64, H - 1, H // 2})
    want = mo.latent_attention(q[:, heads].float().numpy(), c_kv.float().numpy(), k_rope.float().numpy(), cos, sin,
                               ctx, 0.07)
    np.testing.assert_allclose(wide[:, heads], want, **TOL[dtype])


def test_wide_kernel_scores_that_keep_growing(ops, device):
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
    out = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), torch.from_numpy(cos).to(device),
                         torch.from_numpy(sin).to(device), torch.from_numpy(ctx).to(device), S, scale)
    np.testing.assert_allclose(out.cpu().numpy(), want, **TOL[torch.bfloat16])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 128, 700, 16), (1, 100, 3000, 64), (3, 128, 130, 1), (2, 72, 1500, 8), (2, 128, 333, 128)])
def test_wide_kernel_paged_equals_dense(ops, device, shape, dtype):
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
    dense = ops.mla_decode(q.to(device), c_kv.to(device), k_rope.to(device), cos_d, sin_d, ctx_d, max_ctx, 0.11)
    pool_d = pool.to(device)
    paged = ops.mla_decode_paged(q.to(device), pool_d[..., :R], pool_d[..., R:], torch.from_numpy(bt).to(device),
                                 cos_d, sin_d, ctx_d, max_ctx, 0.11)
    torch.cuda.synchronize()
    assert not torch.isnan(paged).any()
    assert torch.equal(paged.view(torch.int32), dense.view(torch.int32))
