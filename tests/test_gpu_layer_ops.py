"""Fused layer steps around the attention kernels (RoPE + KV write, RMSNorm, SwiGLU gating)
against the torch expressions of the reference model (05_mini_vllm/mini_vllm.py:344-357,
:415-427, :490-499), and the HIP-graph decode step of the engine against its eager path.

Tolerances: fp32 kernels follow the reference's operation order, atol 1e-6 (rtol 1e-5: the
row mean is summed in a different order); 16-bit kernels compute in fp32 and round once, so
they are compared with the fp32 expression on the same rounded inputs at 1 ulp of the output
dtype (bf16 rtol 8e-3, fp16 rtol 1e-3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(atol=1e-6, rtol=1e-5), torch.bfloat16: dict(atol=1e-2, rtol=8e-3),
       torch.float16: dict(atol=1e-3, rtol=1e-3)}


@pytest.fixture(scope='module')
def ops(device):
    from mi355_paged import ops
    return ops


def rotate_half(x):
    lo, hi = x.chunk(2, dim=-1)
    return torch.cat((-hi, lo), dim=-1)


def rope_tables(D, n, dtype, device):
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.outer(torch.arange(n).float(), inv)
    tab = torch.cat([ang, ang], -1)
    return tab.cos().to(dtype).to(device), tab.sin().to(dtype).to(device)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('T,Hq,Hkv,D,BS', [(1, 4, 4, 16, 8), (37, 8, 8, 64, 16), (5, 8, 2, 128, 4), (300, 4, 4, 32, 16)])
def test_rope_kv_write_matches_torch(ops, device, dtype, T, Hq, Hkv, D, BS):
    g = torch.Generator(device='cpu').manual_seed(T * 131 + D)
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, generator=g).to(dtype).to(device)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    q, k, v = q.view(T, Hq, D), k.view(T, Hkv, D), v.view(T, Hkv, D)
    cos, sin = rope_tables(D, 512, dtype, device)
    pos = torch.randint(0, 512, (T,), generator=g).to(device)
    NB = (T + BS - 1) // BS + 3
    slots = torch.randperm(NB * BS, generator=g)[:T].to(device)
    if T > 4:
        slots[3] = -1                                          # skipped row
    kc = torch.zeros(NB, BS, Hkv, D, dtype=dtype, device=device)
    vc = torch.zeros_like(kc)
    c, s = cos[pos].float().unsqueeze(1), sin[pos].float().unsqueeze(1)
    want_q = q.float() * c + rotate_half(q.float()) * s
    want_k = k.float() * c + rotate_half(k.float()) * s
    v0 = v.clone()
    ops.rope_kv_write(q, k, v, cos, sin, pos, slots, kc, vc, layout='NSHD')
    tol = TOL[dtype]
    torch.testing.assert_close(q.float(), want_q, **tol)
    torch.testing.assert_close(k.float(), want_k, **tol)
    assert torch.equal(v, v0)                                  # v is only read
    flat_k, flat_v = kc.view(NB * BS, Hkv, D), vc.view(NB * BS, Hkv, D)
    keep = slots >= 0
    assert torch.equal(flat_k[slots[keep]], k[keep])            # the cache holds exactly the rotated rows
    assert torch.equal(flat_v[slots[keep]], v[keep])
    untouched = torch.ones(NB * BS, dtype=torch.bool, device=device)
    untouched[slots[keep]] = False
    assert not flat_k[untouched].any() and not flat_v[untouched].any()


def test_rope_only_and_other_cache_layout(ops, device):
    T, H, D, BS, NB = 9, 4, 32, 4, 5
    q = torch.randn(T, H, D, device=device)
    k = torch.randn(T, H, D, device=device)
    v = torch.randn(T, H, D, device=device)
    cos, sin = rope_tables(D, 64, torch.float32, device)
    pos = torch.arange(T, device=device)
    q0, k0 = q.clone(), k.clone()
    ops.rope_kv_write(q, k, None, cos, sin, pos)               # rotate only
    c, s = cos[pos].unsqueeze(1), sin[pos].unsqueeze(1)
    torch.testing.assert_close(q, q0 * c + rotate_half(q0) * s, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(k, k0 * c + rotate_half(k0) * s, atol=1e-6, rtol=1e-6)
    # [blocks, heads, block_size, D] caches (the KVCachePool layout)
    kc = torch.zeros(NB, H, BS, D, device=device)
    vc = torch.zeros_like(kc)
    slots = torch.arange(T, device=device) + 2
    k1 = k.clone()
    ops.rope_kv_write(q, k, v, cos, sin, pos, slots, kc, vc, layout='NHSD')
    for t in range(T):
        sl = int(slots[t])
        assert torch.equal(kc[sl // BS, :, sl % BS], k[t]) and torch.equal(vc[sl // BS, :, sl % BS], v[t])
    assert not torch.equal(k, k1)                              # rotated a second time in place


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('T,hidden', [(1, 64), (7, 512), (130, 1536), (3, 4096)])
def test_rmsnorm_matches_torch(ops, device, dtype, T, hidden):
    g = torch.Generator().manual_seed(hidden + T)
    x = (torch.randn(T, hidden, generator=g) * 3).to(dtype).to(device)
    d = torch.randn(T, hidden, generator=g).to(dtype).to(device)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).to(device)
    eps = 1e-6

    def ref(t):
        t = t.float()
        return w.float() * (t / (t * t).mean(-1, keepdim=True).add(eps).sqrt())

    tol = TOL[dtype]
    torch.testing.assert_close(ops.rmsnorm(x, w, eps).float(), ref(x), **tol)
    total, normed = ops.rmsnorm(x, w, eps, delta=d)
    want_total = (x.float() + d.float()).to(dtype)
    assert torch.equal(total, want_total)
    torch.testing.assert_close(normed.float(), ref(want_total), **tol)
    # strided input rows (a column slice of a wider activation)
    wide = torch.randn(T, hidden * 2, generator=g).to(dtype).to(device)
    torch.testing.assert_close(ops.rmsnorm(wide[:, :hidden], w, eps).float(), ref(wide[:, :hidden]), **tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_silu_mul_matches_torch(ops, device, dtype):
    g = torch.Generator().manual_seed(3)
    both = (torch.randn(33, 2 * 1536, generator=g) * 4).to(dtype).to(device)
    gate, up = both[:, :1536], both[:, 1536:]
    want = torch.nn.functional.silu(gate.float()) * up.float()
    torch.testing.assert_close(ops.silu_mul(gate, up).float(), want, **TOL[dtype])


def test_layer_ops_reject_bad_arguments(ops, device):
    x = torch.zeros(2, 12, device=device)
    from mi355_paged._lib import Mi355Error
    with pytest.raises(Mi355Error, match='multiple'):
        ops.rmsnorm(torch.zeros(2, 6, device=device), torch.ones(6, device=device), 1e-6)
    with pytest.raises(ValueError):
        ops.silu_mul(x, x[:, :8])
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.rmsnorm(torch.zeros(2, 8), torch.ones(8), 1e-6)


# ------------------------------------------------------------------ engine: graph replay == eager
def _run(mv, graphs, dtype, prompts, max_new, temperature=0.0, seed=0, vocab=300, **kw):
    torch.manual_seed(7)
    eng = mv.MiniVLLM(vocab_size=vocab, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                      num_kv_blocks=256, block_size=8, max_num_seqs=16, max_num_batched_tokens=256,
                      device='cuda', dtype=dtype, decode_graphs=graphs, **kw)
    for p, n in zip(prompts, max_new):
        eng.add_request(p, max_new_tokens=n, temperature=temperature)
    logits = []
    forward = eng.model.forward_planned
    eng.model.forward_planned = lambda *a, **k: logits.append(forward(*a, **k)) or logits[-1]
    torch.manual_seed(seed)
    outs, steps = {}, 0
    while eng.scheduler.has_unfinished:
        for rid, tok in eng.step():
            outs.setdefault(rid, []).append(tok)
        steps += 1
    return eng, outs, steps


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_engine_graph_decode_equals_eager(device, dtype):
    """Same weights and prompts: the engine that replays captured decode steps emits exactly the
    token ids of the engine that runs every step eagerly (greedy), over a run in which the batch
    shrinks as requests finish at different lengths and block tables grow across block borders."""
    import mini_vllm as mv
    rng = np.random.default_rng(11)
    prompts = [[int(t) for t in rng.integers(1, 300, size=int(n))] for n in (5, 17, 9, 30, 2, 12, 8, 21, 3, 15, 6)]
    max_new = [12, 30, 7, 19, 25, 4, 16, 9, 30, 11, 22]
    eager, out_e, steps_e = _run(mv, False, dtype, prompts, max_new)
    graph, out_g, steps_g = _run(mv, True, dtype, prompts, max_new)
    assert not eager._graphs and len(graph._graphs) >= 2         # several batch-size buckets were captured
    assert steps_e == steps_g
    assert out_e == out_g
    assert graph.allocator.num_free == 256 and eager.allocator.num_free == 256
    assert [r.request_id for r in graph.scheduler.finished] == [r.request_id for r in eager.scheduler.finished]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('graphs', [False, True])
def test_engine_fused_decode_step_equals_rope_write_then_attention(device, dtype, graphs):
    """A layer's decode branch as ONE launch (RoPE of q and k, cache write and attention:
    mi355_paged_decode_append_rope) against the two launches it replaces: the same logits, bit for
    bit, at every step, the same tokens, and the same cache contents at the end of the run."""
    import mini_vllm as mv
    rng = np.random.default_rng(41)
    prompts = [[int(t) for t in rng.integers(1, 300, size=int(n))] for n in (5, 17, 9, 30, 2, 12, 8, 21, 3)]
    max_new = [12, 30, 7, 19, 25, 4, 16, 9, 30]
    calls = {'fused': 0}
    from mi355_paged import ops as _ops
    inner = _ops.paged_decode_append

    def counting(*a, **k):
        calls['fused'] += 1 if k.get('rope') is not None else 0
        return inner(*a, **k)

    _ops.paged_decode_append = counting
    mixed_supported = mv.MiniVLLM._mixed_supported
    if not graphs:      # the planned eager path (steps at their own sizes) instead of the bucketed step bodies
        mv.MiniVLLM._mixed_supported = lambda self: False
    try:
        two, out_two, steps_two = _run(mv, graphs, dtype, prompts, max_new, fused_decode_step=False, decode_lookahead=False)
        assert calls['fused'] == 0
        one, out_one, steps_one = _run(mv, graphs, dtype, prompts, max_new, fused_decode_step=True, decode_lookahead=False)
        assert calls['fused'] > 0, "the fused launch was never taken"
    finally:
        _ops.paged_decode_append = inner
        mv.MiniVLLM._mixed_supported = mixed_supported
    assert out_one == out_two and steps_one == steps_two
    assert torch.equal(one.model.k_cache.view(torch.uint8), two.model.k_cache.view(torch.uint8))
    assert torch.equal(one.model.v_cache.view(torch.uint8), two.model.v_cache.view(torch.uint8))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_engine_lookahead_decode_equals_step_by_step(device, dtype):
    """The engine that issues the next decode step before reading the current one's tokens emits,
    per request, exactly the tokens of the step-by-step engines -- over a run in which requests end
    on the end-of-sequence id (vocabulary of 12: id 0 is sampled often) at steps nobody can predict,
    others end by length, and the batch shrinks accordingly."""
    import mini_vllm as mv
    rng = np.random.default_rng(23)
    prompts = [[int(t) for t in rng.integers(1, 12, size=int(n))] for n in rng.integers(2, 40, size=16)]
    max_new = [int(n) for n in rng.integers(3, 60, size=16)]
    eager, out_e, _ = _run(mv, False, dtype, prompts, max_new, vocab=12)
    plain, out_p, steps_p = _run(mv, True, dtype, prompts, max_new, vocab=12, decode_lookahead=False)
    ahead, out_a, steps_a = _run(mv, True, dtype, prompts, max_new, vocab=12, decode_lookahead=True)
    ended_early = sum(1 for rid, toks in out_e.items() if toks[-1] == 0 and len(toks) < max_new[int(rid.split('-')[1])])
    assert ended_early >= 3, "the case is meant to exercise end-of-sequence ids"
    assert out_p == out_e
    assert out_a == out_e
    assert steps_a == steps_p
    assert ahead._inflight is None
    assert ahead.allocator.num_free == 256
    assert sorted(r.request_id for r in ahead.scheduler.finished) == sorted(r.request_id for r in eager.scheduler.finished)


def test_engine_lookahead_eos_on_a_block_boundary_with_prefix_caching(device):
    """A request that samples the end-of-sequence id has the next step already issued; that
    step's row writes K/V one position further -- into a block taken early when the position
    opens a new block -- and the request's blocks are freed right after.  With prefix caching on,
    freed prompt blocks stay adoptable: the stray row must never land in one of them, and a freed
    block handed to another request must end up holding that request's rows.  The case is searched
    for, not hoped for: prompts share block-aligned prefixes, the pool is small enough that freed
    blocks are reused at once, and the run must contain requests that end on the end-of-sequence
    id with their next position on a block boundary.  Two passes over the same prompts: the second
    adopts the blocks the first left cached, and both must give the eager engine's tokens."""
    import mini_vllm as mv
    bs = 8

    def workload(seed):
        rng = np.random.default_rng(seed)
        stems = [[int(t) for t in rng.integers(1, 12, size=n)] for n in (16, 24, 8)]
        prompts = []
        for i in range(48):
            tail = [int(t) for t in rng.integers(1, 12, size=int(rng.integers(1, 9)))]
            prompts.append(stems[i % 3] + tail)
        return prompts, [int(n) for n in rng.integers(6, 40, size=len(prompts))]

    def run(prompts, max_new, **kw):
        torch.manual_seed(7)
        eng = mv.MiniVLLM(vocab_size=12, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                          num_kv_blocks=48, block_size=bs, max_num_seqs=6, max_num_batched_tokens=256,
                          device='cuda', dtype=torch.float32, **kw)
        passes = []
        for _ in range(2):
            ids = [eng.add_request(p, max_new_tokens=n, temperature=0.0) for p, n in zip(prompts, max_new)]
            outs = {rid: [] for rid in ids}
            while eng.scheduler.has_unfinished:
                for rid, tok in eng.step():
                    outs[rid].append(tok)
            passes.append([outs[rid] for rid in ids])
        return eng, passes

    # the k-th output being the end-of-sequence id, the step issued ahead feeds it at position
    # prompt_len + k - 1: that row's cache write opens a new block when the position is a multiple of bs
    for seed in range(5, 15):
        prompts, max_new = workload(seed)
        _, want = run(prompts, max_new, decode_graphs=False)
        boundary_ends = [i for i, toks in enumerate(want[0])
                         if toks[-1] == 0 and len(toks) < max_new[i] and (len(prompts[i]) + len(toks) - 1) % bs == 0]
        if len(boundary_ends) >= 2:
            break
    assert len(boundary_ends) >= 2, "the case is meant to end requests with their next position opening a block"
    assert want[0] == want[1]                       # greedy: a request's tokens depend on its prompt only
    eng, got = run(prompts, max_new, decode_graphs=True, decode_lookahead=True, enable_prefix_caching=True)
    assert got == want
    assert eng._launches > 20 and eng._inflight is None
    assert eng.allocator.num_free == 48
    cached = run(prompts, max_new, decode_graphs=False, enable_prefix_caching=True)[1]
    assert cached == want


def test_engine_follows_weight_edits_and_replacements(device):
    """Weights changed after steps have been captured: written in place through `.data` (which does
    not bump a tensor's version counter), and replaced by new tensors.  Both the fused gate/up
    projection and the captured decode steps must follow -- the results are those of an engine
    built with the edited weights from the start."""
    import mini_vllm as mv
    prompts = [[3, 5, 7, 11, 13], [2, 4, 6], [9, 8, 7, 6, 5, 4, 3, 2, 1]]

    def build():
        torch.manual_seed(21)
        eng = mv.MiniVLLM(vocab_size=300, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                          num_kv_blocks=64, block_size=8, max_num_seqs=8, max_num_batched_tokens=128,
                          device='cuda', dtype=torch.float32)
        for p in eng.model.parameters():
            p.requires_grad_(False)
        return eng

    def gen(eng):
        ids = [eng.add_request(p, max_new_tokens=10, temperature=0.0) for p in prompts]
        outs = {rid: [] for rid in ids}
        while eng.scheduler.has_unfinished:
            for rid, tok in eng.step():
                outs[rid].append(tok)
        return [outs[r] for r in ids]

    def edit_in_place(eng):
        g = torch.Generator(device='cpu').manual_seed(5)
        ffn = eng.model.layers[0].ffn
        ffn.gate.weight.data.copy_(torch.randn(ffn.gate.weight.shape, generator=g).to(device) * 0.2)
        ffn.up.weight.data.mul_(-1.5)
        eng.model.lm_head.weight.data.add_(torch.randn(eng.model.lm_head.weight.shape, generator=g).to(device) * 0.05)

    def replace(eng):
        g = torch.Generator(device='cpu').manual_seed(6)
        ffn = eng.model.layers[1].ffn
        ffn.up.weight = torch.nn.Parameter(torch.randn(ffn.up.weight.shape, generator=g).to(device) * 0.1,
                                           requires_grad=False)
        eng.model.lm_head.weight = torch.nn.Parameter(
            torch.randn(eng.model.lm_head.weight.shape, generator=g).to(device) * 0.1, requires_grad=False)

    eng = build()
    base = gen(eng)
    assert len(eng._graphs) >= 1
    edit_in_place(eng)
    after_edit = gen(eng)
    replace(eng)
    after_replace = gen(eng)

    fresh = build()
    edit_in_place(fresh)
    assert gen(fresh) == after_edit and after_edit != base
    fresh = build()
    edit_in_place(fresh)
    replace(fresh)
    assert gen(fresh) == after_replace and after_replace != after_edit


def test_engine_more_requests_than_slots_equals_eager(device):
    """Continuous batching with a waiting queue: captured decode steps (and steps issued ahead) run
    while every sequence slot is taken, the general path takes over the moment a slot frees.  Every
    request gets exactly the tokens the eager engine gives it (greedy; a 12-id vocabulary makes
    end-of-sequence ids frequent, which is when the lookahead engine admits a waiting request one
    step later than the step-by-step engines)."""
    import mini_vllm as mv
    rng = np.random.default_rng(31)
    prompts = [[int(t) for t in rng.integers(1, 12, size=int(n))] for n in rng.integers(2, 30, size=40)]
    max_new = [int(n) for n in rng.integers(3, 40, size=40)]
    eager, out_e, _ = _run(mv, False, torch.float32, prompts, max_new, vocab=12)
    plain, out_p, steps_p = _run(mv, True, torch.float32, prompts, max_new, vocab=12, decode_lookahead=False)
    ahead, out_a, _ = _run(mv, True, torch.float32, prompts, max_new, vocab=12, decode_lookahead=True)
    assert out_p == out_e and out_a == out_e
    assert len(plain._graphs) >= 1 and plain._launches > 10          # the captured path did serve steps
    assert ahead._launches > 10 and ahead._inflight is None
    assert ahead.allocator.num_free == 256 and plain.allocator.num_free == 256


@pytest.mark.parametrize('prefix', [False, True])
def test_engine_steps_with_prompt_chunks_issued_ahead_equal_eager(device, prefix):
    """Prompts longer than the token budget (several chunks each), more requests than slots, a
    12-id vocabulary (frequent end-of-sequence ids): steps that hold prompt chunks are issued before
    the previous step's ids are read, decode rows feeding those ids from device memory.  Every
    request gets the tokens the step-by-step engines give it; a request that drew the end-of-sequence
    id costs one dropped row; blocks all return."""
    import mini_vllm as mv
    rng = np.random.default_rng(77)
    prompts = [[int(t) for t in rng.integers(1, 12, size=int(n))] for n in rng.integers(3, 90, size=36)]
    if prefix:
        for p in prompts[::3]:
            p[:16] = prompts[0][:16] if len(prompts[0]) >= 16 else p[:16]
    max_new = [int(n) for n in rng.integers(2, 24, size=36)]

    def run(**kw):
        torch.manual_seed(3)
        eng = mv.MiniVLLM(vocab_size=12, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                          num_kv_blocks=160, block_size=8, max_num_seqs=6, max_num_batched_tokens=40,
                          device='cuda', dtype=torch.float32, enable_prefix_caching=prefix, **kw)
        ids = [eng.add_request(p, max_new_tokens=n, temperature=0.0) for p, n in zip(prompts, max_new)]
        outs = {rid: [] for rid in ids}
        steps = 0
        while eng.scheduler.has_unfinished:
            for rid, tok in eng.step():
                outs[rid].append(tok)
            steps += 1
        assert eng.allocator.num_free == 160 and eng._flight is None and eng._deferred is None
        return eng, [outs[r] for r in ids], steps

    eager, want, _ = run(decode_graphs=False)
    assert all(len(t) >= 1 and all(0 <= x < 12 for x in t) for t in want)
    assert sum(1 for t, n in zip(want, max_new) if t[-1] == 0 and len(t) < n) >= 5
    stepwise, got_s, steps_s = run(pipeline_steps=False)
    ahead, got_a, steps_a = run(pipeline_steps=True)
    assert got_s == want and got_a == want
    assert stepwise._ahead == 0 and ahead._ahead > 30
    assert ahead.scheduler.dropped_items >= 1 and stepwise.scheduler.dropped_items == 0
    assert steps_a <= steps_s + ahead.scheduler.dropped_items + 2


def test_engine_step_scheduled_ahead_that_cannot_be_issued_ahead(device, monkeypatch):
    """A step scheduled while its predecessor is in flight, but whose contexts the replayable body
    does not serve (here: the staged-table limit lowered to 256 tokens), waits for the general path:
    by then its decode items carry real ids and requests that ended meanwhile are dropped."""
    import mini_vllm as mv
    from mi355_paged import ops as _ops
    monkeypatch.setattr(_ops, 'paged_prefill_serves', lambda dtype, head_dim, bs, n_q, context: context <= 32 * bs)
    rng = np.random.default_rng(5)
    lens = [300, 20, 12, 330, 9, 40, 280, 15, 22, 310, 7, 18]
    prompts = [[int(t) for t in rng.integers(1, 12, size=n)] for n in lens]
    max_new = [int(n) for n in rng.integers(3, 20, size=len(lens))]

    def run(**kw):
        torch.manual_seed(9)
        eng = mv.MiniVLLM(vocab_size=12, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                          num_kv_blocks=256, block_size=8, max_num_seqs=4, max_num_batched_tokens=48,
                          device='cuda', dtype=torch.float32, **kw)
        ids = [eng.add_request(p, max_new_tokens=n, temperature=0.0) for p, n in zip(prompts, max_new)]
        outs = {rid: [] for rid in ids}
        while eng.scheduler.has_unfinished:
            for rid, tok in eng.step():
                outs[rid].append(tok)
        assert eng.allocator.num_free == 256 and eng._flight is None and eng._deferred is None
        return eng, [outs[r] for r in ids]

    _, want = run(decode_graphs=False)
    eng, got = run()
    assert got == want
    assert eng._deferrals >= 2 and eng._ahead >= 4


def test_engine_lookahead_accepts_requests_mid_run(device):
    """A request added while a speculative step is in flight joins through the general path and
    every request still gets the tokens it gets when run alone (greedy)."""
    import mini_vllm as mv

    def build():
        torch.manual_seed(7)
        return mv.MiniVLLM(vocab_size=300, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256,
                           num_kv_blocks=256, block_size=8, max_num_seqs=16, max_num_batched_tokens=256,
                           device='cuda', dtype=torch.float32)
    first, late = [5, 6, 7, 8, 9], [11, 12, 13]
    eng = build()
    a = eng.add_request(first, max_new_tokens=20, temperature=0.0)
    outs = {}
    for _ in range(6):
        for rid, tok in eng.step():
            outs.setdefault(rid, []).append(tok)
    assert eng._inflight is not None
    b = eng.add_request(late, max_new_tokens=9, temperature=0.0)
    while eng.scheduler.has_unfinished:
        for rid, tok in eng.step():
            outs.setdefault(rid, []).append(tok)
    def alone(prompt, n):
        solo = build()
        solo.add_request(prompt, max_new_tokens=n, temperature=0.0)
        toks = []
        while solo.scheduler.has_unfinished:
            toks += [tok for _, tok in solo.step()]
        return toks
    assert outs[a] == alone(first, 20)
    assert outs[b] == alone(late, 9)


def test_engine_graph_decode_stochastic_rows_repeat(device):
    """Sampling after a replayed step draws from the seeded generator like the eager path."""
    import mini_vllm as mv
    prompts = [[3, 4, 5], [9, 8, 7, 6], [1, 2]]
    _, a, _ = _run(mv, True, torch.float32, prompts, [10, 10, 10], temperature=0.9, seed=5)
    _, b, _ = _run(mv, False, torch.float32, prompts, [10, 10, 10], temperature=0.9, seed=5)
    assert a == b


def test_decode_step_inputs_bit_exact(ops, device):
    """Slot mapping, context lengths and gathered block tables of a decode batch, integer-exact
    against the host formulas (05_mini_vllm/mini_vllm.py:209-219), padding rows included."""
    rng = np.random.default_rng(5)
    pool_rows, width, BS, rows = 40, 9, 16, 24
    pool = rng.integers(0, 5000, size=(pool_rows, width)).astype(np.int32)
    tok = rng.integers(0, 32000, size=rows).astype(np.int32)
    pos = rng.integers(0, width * BS, size=rows).astype(np.int32)
    row = rng.integers(0, pool_rows, size=rows).astype(np.int32)
    pos[[3, 17]] = -1                                            # padding rows
    packed = torch.from_numpy(np.stack([tok, pos, row])).to(device)
    t, p, s, c, bt = ops.decode_step_inputs(packed, torch.from_numpy(pool).to(device), BS)
    assert t.tolist() == tok.tolist()
    assert p.tolist() == np.maximum(pos, 0).tolist()
    assert c.tolist() == (pos + 1).clip(min=0).tolist()
    assert bt.cpu().numpy().tolist() == pool[row].tolist()
    want = [-1 if q < 0 else int(pool[r][q // BS]) * BS + q % BS for q, r in zip(pos.tolist(), row.tolist())]
    assert s.tolist() == want
    # token entries -1-j are taken from the previous step's id vector on the device
    prev = torch.from_numpy(rng.integers(0, 32000, size=rows + 5).astype(np.int32)).to(device)
    refs = tok.copy()
    refs[::3] = -1 - rng.integers(0, rows + 5, size=len(refs[::3])).astype(np.int32)
    packed_refs = torch.from_numpy(np.stack([refs, pos, row])).to(device)
    t2, p2, s2, c2, bt2 = ops.decode_step_inputs(packed_refs, torch.from_numpy(pool).to(device), BS, prev_ids=prev)
    prev_host = prev.cpu().numpy()
    assert t2.tolist() == [int(prev_host[-1 - x]) if x < 0 else int(x) for x in refs.tolist()]
    assert s2.tolist() == want and c2.tolist() == c.tolist()
    # a block size that is not a power of two; a position past the table's last block gets no slot
    t, p, s, c, bt = ops.decode_step_inputs(packed, torch.from_numpy(pool).to(device), 12)
    assert s.tolist() == [-1 if q < 0 or q // 12 >= width else int(pool[r][q // 12]) * 12 + q % 12
                          for q, r in zip(pos.tolist(), row.tolist())]


def test_captured_decode_keeps_its_own_workspace(ops, device):
    """A split-KV decode captured in a HIP graph must not depend on the shared scratch tensor: that
    tensor is replaced when a later call needs more, and the graph would then write into freed memory."""
    rng = np.random.default_rng(2)
    B, H, D, BS, S = 2, 8, 64, 16, 4096                    # 2 sequences: tokens are split, a workspace is needed
    per = S // BS
    NB = B * per + 3
    kc = torch.from_numpy(rng.standard_normal((NB, H, BS, D), dtype=np.float32)).to(torch.bfloat16).to(device)
    vc = torch.from_numpy(rng.standard_normal((NB, H, BS, D), dtype=np.float32)).to(torch.bfloat16).to(device)
    q = torch.from_numpy(rng.standard_normal((B, H, D), dtype=np.float32)).to(torch.bfloat16).to(device)
    bt = torch.from_numpy(rng.permutation(NB)[:B * per].reshape(B, per).astype(np.int32)).to(device)
    ctx = torch.tensor([S, S - 77], dtype=torch.int32, device=device)
    want = ops.paged_decode(q, kc, vc, bt, ctx, S).clone()
    out = torch.empty_like(want)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.paged_decode(q, kc, vc, bt, ctx, S, out=out)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.paged_decode(q, kc, vc, bt, ctx, S, out=out)
    # force the shared scratch to be replaced by a much larger one, and scribble over fresh allocations
    need = ops._workspaces[device].numel() if device in ops._workspaces else 0
    big_q = torch.zeros((64, 64, D), dtype=torch.bfloat16, device=device)
    big_bt = bt[:1].repeat(64, 1)
    ops.paged_decode(big_q, kc, vc, big_bt, torch.full((64,), 64, dtype=torch.int32, device=device), S)
    ops._workspaces.pop(device, None)
    junk = [torch.full((1 << 20,), 1e30, device=device) for _ in range(8)]
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and need >= 0 and len(junk) == 8


def test_native_rccl_communicator_one_rank(device):
    """mi355_dp_*: unique id, communicator, all-gather and teardown on a one-rank RCCL communicator
    (a one-GPU machine cannot hold more ranks); through the overlapped gather as well."""
    from mi355_paged import dp
    comm = dp.NativeComm(device)
    assert (comm.rank, comm.world) == (0, 1)
    send = torch.arange(100, 356, dtype=torch.int32, device=device)
    recv = torch.zeros(256, dtype=torch.int32, device=device)
    comm.all_gather_i32(send, recv)
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    with pytest.raises(ValueError):
        comm.all_gather_i32(send, recv[:100])
    g = dp.OverlappedTokenGather(8, device, single_rank_collective=True, every=3, native=comm)
    for i in range(10):
        g.send_buffer(i).copy_(torch.arange(8, dtype=torch.int32, device=device) + 10 * i)
        g.launch(i)
    g.drain()
    assert g.result(9).tolist() == [90 + j for j in range(8)]
    assert g.result(7).tolist() == [70 + j for j in range(8)]
    comm.close()


def test_exchange_started_by_the_gpu_one_rank(ops, device):
    """mi355_dp_exchange_* on a one-rank RCCL communicator: each all-gather sits on the library's side
    stream behind a watcher kernel that returns once the step's ids are written; the compute stream
    only runs the plain sampler.  The ids of step i must be in result(i) -- never those of an earlier
    or later step -- over 50 steps whose logits change every step, with a long kernel queued ahead
    so that the host runs far ahead of the device (and is held back by the buffer ring, not the
    stream); afterwards every send buffer is marked "not written" again."""
    from mi355_paged import dp
    comm = dp.NativeComm(device)
    B, V = 256, 4096
    gather = dp.OverlappedTokenGather(B, device, single_rank_collective=True, native=comm)
    assert gather.watcher
    assert all(bool((s == -1).all()) for s in gather.send)
    g = torch.Generator(device='cpu').manual_seed(5)
    logits = [torch.randn((B, V), generator=g).to(device) for _ in range(5)]
    want = [l.argmax(dim=1).to(torch.int32) for l in logits]
    torch.cuda._sleep(20_000_000)
    kept = {}
    for i in range(50):
        ops.sample_argmax(logits[i % 5], out=gather.send_buffer(i))
        gather.launch(i)
        if i >= 2:      # results stay valid for SLOTS - 1 further steps
            kept[i - 2] = gather.result(i - 2).clone()
    gather.drain()
    kept[48], kept[49] = gather.result(48).clone(), gather.result(49).clone()
    torch.cuda.synchronize()
    for i in range(50):
        assert torch.equal(kept[i], want[i % 5]), i
    assert all(bool((s == -1).all()) for s in gather.send)
    # ids written by a kernel of another library (a torch copy) start the exchange just the same
    for i in range(50, 60):
        gather.send_buffer(i).copy_(want[i % 5] + i)
        gather.launch(i)
    gather.drain()
    assert torch.equal(gather.result(59), want[59 % 5] + 59) and torch.equal(gather.result(57), want[57 % 5] + 57)
    # the form with one event record (no system fence) per exchange on the compute stream
    plain = dp.OverlappedTokenGather(B, device, single_rank_collective=True, native=comm, device_started=False)
    assert not plain.watcher
    ops.sample_argmax(logits[0], out=plain.send_buffer(0))
    plain.launch(0)
    plain.drain()
    assert torch.equal(plain.result(0), want[0])
    gather.close()
    comm.close()


def test_exchange_fed_from_host_vectors_one_rank(device):
    """OverlappedTokenGather.host_buffer / launch_from_host with the library's communicator: the
    upload and the collective replay as one graph per buffer pair on the side stream."""
    from mi355_paged import dp
    comm = dp.NativeComm(device)
    gather = dp.OverlappedTokenGather(33, device, single_rank_collective=True, native=comm)
    for i in range(21):
        host = gather.host_buffer(i)
        host.copy_(torch.arange(33, dtype=torch.int32) + 100 * i)
        gather.launch_from_host(i)
        if i >= 1:
            assert gather.result(i - 1).cpu().tolist() == [100 * (i - 1) + j for j in range(33)]
    gather.drain()
    assert gather.result(20).cpu().tolist() == [2000 + j for j in range(33)]
    assert gather._host_graphs, "the exchange was not captured"
    comm.close()
