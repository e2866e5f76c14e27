"""The CPU oracle against the golden vectors captured from the reference (CPU only)."""
import numpy as np
import pytest
import torch

from conftest import load_json, load_npz, npz_cases
from oracle import loop_form, paged_oracle as orc


def test_g1_known_answer():
    z = load_npz('g1_decode_known_answer.npz')
    cache = z['kv_cache']
    out = orc.paged_decode(z['query'], cache[0], cache[1], z['block_table'].tolist(),
                           int(z['seq_len']), int(z['block_size']))
    # same bar as the reference's own test (01_paged_attention/test_paged_attention.py:99)
    np.testing.assert_allclose(out, z['out_paged'], atol=1e-5, rtol=0)
    np.testing.assert_allclose(out, z['out_dense'], atol=1e-5, rtol=0)


@pytest.mark.parametrize('case', npz_cases('g2_decode_cases.npz'), ids=lambda c: str(c[0]))
def test_g2_decode_cases(case):
    m, f = case
    cache = f['kv_cache'].astype(np.float32)
    q = f['query'].astype(np.float32)
    bt = f['block_table'].tolist()
    out = orc.paged_decode(q, cache[0], cache[1], bt, m['s'], m['bs'])
    np.testing.assert_allclose(out, f['out'], atol=1e-5, rtol=1e-5)
    out64 = orc.paged_decode(q, cache[0], cache[1], bt, m['s'], m['bs'], dtype=np.float64)
    np.testing.assert_allclose(out64, f['out'], atol=1e-5, rtol=1e-5)
    loop = loop_form.decode_one_sequence(torch.from_numpy(q), torch.from_numpy(cache), bt, m['s'], m['bs'])
    np.testing.assert_allclose(loop.numpy(), f['out'], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize('case', npz_cases('g3_prefill_cases.npz'), ids=lambda c: str(c[0]))
def test_g3_prefill_cases(case):
    m, f = case
    cache = f['cache_before'].astype(np.float32).copy()
    out = orc.paged_prefill(f['query'].astype(np.float32), f['key'].astype(np.float32),
                            f['value'].astype(np.float32), cache[0], cache[1],
                            f['block_table'].tolist(), m['bs'])
    np.testing.assert_allclose(out, f['out'], atol=1e-5, rtol=1e-5)
    np.testing.assert_array_equal(cache, f['cache_after'].astype(np.float32))   # writes are exact copies


def test_g4_slot_mapping():
    for c in load_json('g4_slot_mapping.json'):
        assert orc.slot_mapping(c['block_table'], c['seq_len'], c['block_size']) == c['slots']
        assert orc.slot_mapping_range(c['block_table'], 0, c['seq_len'], c['block_size']) == c['slots']


def test_g5_pool_model():
    g = load_json('g5_allocator_traces.json')['pool']
    pool = orc.PoolModel(g['num_blocks'])
    for rec in g['trace']:
        if rec['op'] == 'alloc':
            if 'error' in rec:
                with pytest.raises(RuntimeError, match='OOM'):
                    pool.allocate(rec['arg'])
            else:
                assert pool.allocate(rec['arg']) == rec['result']
        else:
            pool.free(rec['arg'])
        assert pool.free_blocks == rec['free_blocks']
        assert pool.ref_counts == rec['ref_counts']


def _row(vals):
    conv = {None: float('nan'), 'inf': float('inf'), '-inf': float('-inf')}
    return [conv[v] if (v is None or isinstance(v, str)) else v for v in vals]


def test_g7_greedy():
    g = load_json('g7_sampler.json')
    for row, want in zip(g['greedy_rows'], g['greedy_ids']):
        assert orc.argmax_rows(np.array([_row(row)], dtype=np.float32)) == [want]
    z = load_npz('g7_sampler_big.npz')
    assert orc.argmax_rows(z['logits']) == z['greedy_ids'].tolist()


def test_g7_nucleus():
    for c in load_json('g7_sampler.json')['nucleus']:
        probs, kept = orc.nucleus_probs(np.array(c['logits'], dtype=np.float32), c['temperature'], c['top_p'])
        ids = c['kept_ids_sorted']
        assert sorted(np.nonzero(kept)[0].tolist()) == sorted(ids)          # kept set exact
        np.testing.assert_allclose(probs[ids], c['kept_probs_sorted'], rtol=1e-5, atol=1e-7)
        assert abs(float(probs.sum()) - 1.0) < 1e-5


def test_g7_nucleus_at_vocab_32000():
    """The kept set the reference's Sampler produces at the benchmark's vocabulary size."""
    z = load_npz('g7_sampler_vocab32000.npz')
    for i in range(len(z['temperature'])):
        probs, kept = orc.nucleus_probs(z['logits'][i], float(z['temperature'][i]), float(z['top_p'][i]))
        ids = z[f'kept_ids_sorted_{i}']
        assert float(z[f'margin_{i}']) > 8          # the cut is not within rounding of top_p
        assert np.array_equal(np.nonzero(kept)[0], np.sort(ids))                  # kept set exact
        np.testing.assert_allclose(probs[ids], z[f'kept_probs_sorted_{i}'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('case', npz_cases('g2_decode_cases.npz'), ids=lambda c: str(c[0]))
def test_c_port_against_golden(case):
    from oracle import c_port
    m, f = case
    cache = f['kv_cache'].astype(np.float32)
    out = c_port.paged_decode(f['query'].astype(np.float32)[None], cache[0], cache[1],
                              f['block_table'][None], [m['s']])[0]
    np.testing.assert_allclose(out, f['out'], atol=1e-5, rtol=1e-5)


def test_c_port_known_answer_and_batch():
    from oracle import c_port
    z = load_npz('g1_decode_known_answer.npz')
    cache = z['kv_cache']
    out = c_port.paged_decode(z['query'][None], cache[0], cache[1], z['block_table'][None], [int(z['seq_len'])])
    np.testing.assert_allclose(out[0], z['out_paged'], atol=1e-5, rtol=0)
    rng = np.random.default_rng(0)
    kc = rng.standard_normal((2, 9, 2, 4, 16), dtype=np.float32)
    q = rng.standard_normal((3, 4, 16), dtype=np.float32)
    bt = np.array([[1, 5, 0], [2, 3, 4], [8, 7, 6]])
    ctx = [12, 0, 5]
    want = orc.paged_decode_batch(q, kc[0], kc[1], bt, ctx, 4)
    np.testing.assert_allclose(c_port.paged_decode(q, kc[0], kc[1], bt, ctx), want, atol=1e-5, rtol=1e-5)


# ------------------------------------------------------------------ MLA (04_mla)
def _mla_models():
    import json
    z = load_npz('g9_mla.npz')
    meta = json.loads(str(z['meta']))
    for ci, m in enumerate(meta['models']):
        w = {k[len(f'm{ci}_w::'):]: z[k].astype(np.float32) for k in z.files if k.startswith(f'm{ci}_w::')}
        yield ci, m['config'], m['calls'], w, z


def test_g9_mla_forward_matches_reference():
    from oracle import mla_oracle as mo
    for ci, cfg, calls, w, z in _mla_models():
        for name in calls:
            pre = f'm{ci}_{name}_'
            past = (z[pre + 'past_c_kv'], z[pre + 'past_k_rope']) if pre + 'past_c_kv' in z.files else (None, None)
            out, nc, nr = mo.mla_forward(w, cfg, z[pre + 'x'], z[pre + 'pos'], *past)
            np.testing.assert_allclose(out, z[pre + 'out'], atol=2e-5, rtol=1e-4)
            np.testing.assert_allclose(nc, z[pre + 'new_c_kv'], atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(nr, z[pre + 'new_k_rope'], atol=1e-5, rtol=1e-5)


def test_g9_absorbed_form_equals_expanded_form():
    """The latent-space formulation the kernel implements reproduces the reference outputs."""
    from oracle import mla_oracle as mo
    for ci, cfg, calls, w, z in _mla_models():
        H, nope, rope, lora, vd = (cfg[k] for k in ('num_heads', 'qk_nope_head_dim', 'qk_rope_head_dim',
                                                    'kv_lora_rank', 'v_head_dim'))
        for name in calls:
            pre = f'm{ci}_{name}_'
            x, pos = z[pre + 'x'].astype(np.float64), z[pre + 'pos']
            W = {k: v.astype(np.float64) for k, v in w.items()}
            B, T, _ = x.shape
            q = mo.rms_norm(x @ W['q_a_proj.weight'].T, W['q_a_layernorm.weight'], eps=np.finfo(np.float32).eps)
            q = (q @ W['q_b_proj.weight'].T).reshape(B, T, H, nope + rope)
            c_kv, k_rope = z[pre + 'new_c_kv'].astype(np.float64), z[pre + 'new_k_rope'].astype(np.float64)
            if pre + 'past_c_kv' in z.files:
                c_kv = np.concatenate([z[pre + 'past_c_kv'], c_kv], 1)
                k_rope = np.concatenate([z[pre + 'past_k_rope'], k_rope], 1)
            S = c_kv.shape[1]
            cos, sin = mo.rope_tables(rope, S + 8, dtype=np.float64)
            q_rope = q[..., nope:] * cos[pos][:, :, None] + mo.rotate_half(q[..., nope:]) * sin[pos][:, :, None]
            q_cat = np.concatenate([mo.absorb_query(q[..., :nope], W['kv_b_proj.weight'], H, nope, vd), q_rope], -1)
            rows = q_cat.reshape(B * T, H, lora + rope)
            ctx = (pos + 1).reshape(-1)
            row_seq = np.repeat(np.arange(B), T)
            o_lat = mo.latent_attention(rows, c_kv, k_rope, cos, sin, ctx, (nope + rope) ** -0.5, row_seq)
            o = mo.expand_output(o_lat, W['kv_b_proj.weight'], H, nope, vd).reshape(B, T, H * vd)
            out = o @ W['o_proj.weight'].T
            np.testing.assert_allclose(out, z[pre + 'out'], atol=2e-5, rtol=1e-4)
