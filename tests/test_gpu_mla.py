"""MLA latent-space attention kernel and the drop-in `mini_mla` module (needs an MI355X).

Tolerances: the kernel takes 16-bit inputs, rotates cached k_rope in fp32 and rounds it and the
softmax weights back to 16 bits before the matrix products, accumulating in fp32 ->
atol=1e-2, rtol=1.6e-2 against the float64 oracle on the same rounded inputs (bf16), and
atol=2e-3, rtol=4e-3 for fp16.  Module outputs with fp32 weights are compared with the
reference module's stored outputs at atol=5e-3 (the attention core runs in fp16)."""
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
    (1, 20, 256, 64, 50), (1, 128, 512, 64, 130), (5, 3, 32, 24, 17),
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
