"""Generate the golden fixtures in this directory from the reference itself.

Run in the build container only (the reference checkout is not present on the
GPU box):

    python tests/golden/make_golden.py [/root/reference]

The script imports the reference's two modules, calls them on seeded inputs and
stores inputs + outputs as small .npz / .json files.  Nothing from the
reference's source text is stored; the fixtures are data.
"""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, '01_paged_attention'))
sys.path.insert(0, os.path.join(REF, '05_mini_vllm'))

import paged_attention as ref_pa  # noqa: E402
import mini_vllm as ref_mv  # noqa: E402


def save_npz(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)


def save_json(name, obj):
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(obj, f, separators=(',', ':'))


# ---------------------------------------------------------------- G1
def g1():
    """Inputs of the reference's own known-answer test
    (01_paged_attention/test_paged_attention.py:67-100)."""
    torch.manual_seed(42)
    H, D, S, BS, NB = 4, 32, 8, 4, 4
    K = torch.randn(S, H, D)
    V = torch.randn(S, H, D)
    Q = torch.randn(H, D)
    scores = torch.einsum('hd,shd->hs', Q, K) * (1.0 / math.sqrt(D))
    dense = torch.einsum('hs,shd->hd', torch.softmax(scores, dim=-1), V)
    cache = torch.zeros(2, NB, H, BS, D)
    bt = [2, 0]
    for i in range(S):
        cache[0, bt[i // BS], :, i % BS] = K[i]
        cache[1, bt[i // BS], :, i % BS] = V[i]
    out = ref_pa.paged_attention_decode(Q, cache, bt, S, D, H, H, block_size=BS)
    assert torch.allclose(dense, out, atol=1e-5)
    save_npz('g1_decode_known_answer.npz', query=Q.numpy(), kv_cache=cache.numpy(),
             block_table=np.array(bt), seq_len=np.array(S), block_size=np.array(BS),
             out_paged=out.numpy(), out_dense=dense.numpy())


# ---------------------------------------------------------------- G2
def g2():
    cases = [
        # Hq, Hkv, D, BS, S
        (8, 8, 64, 16, 1), (8, 8, 64, 16, 15), (8, 8, 64, 16, 16), (8, 8, 64, 16, 17),
        (8, 4, 32, 16, 20), (8, 2, 64, 4, 100), (8, 8, 32, 4, 128), (8, 4, 64, 16, 100),
        (8, 2, 32, 16, 128), (8, 8, 64, 4, 17), (8, 1, 64, 16, 33), (4, 4, 16, 8, 9),
        (8, 8, 128, 16, 40), (6, 2, 32, 16, 21),
    ]
    arrays, meta = {}, []
    g = torch.Generator().manual_seed(1234)
    for i, (hq, hkv, d, bs, s) in enumerate(cases):
        need = (s + bs - 1) // bs
        nb = need + 2
        # values are fp16-representable so the fixture can be stored at half the size
        cache = torch.randn(2, nb, hkv, bs, d, generator=g).half().float()  # unused blocks hold garbage
        q = torch.randn(hq, d, generator=g).half().float()
        bt = torch.randperm(nb, generator=g)[:need].tolist()
        out = ref_pa.paged_attention_decode(q, cache, bt, s, d, hq, hkv, block_size=bs)
        arrays[f'c{i}_query'] = q.numpy().astype(np.float16)
        arrays[f'c{i}_kv_cache'] = cache.numpy().astype(np.float16)
        arrays[f'c{i}_block_table'] = np.array(bt)
        arrays[f'c{i}_out'] = out.numpy()
        meta.append(dict(hq=hq, hkv=hkv, d=d, bs=bs, s=s, nb=nb))
    arrays['meta'] = np.array(json.dumps(meta))
    save_npz('g2_decode_cases.npz', **arrays)


# ---------------------------------------------------------------- G3
def g3():
    cases = [  # T, Hq, Hkv, D, BS
        (1, 4, 4, 16, 4), (4, 4, 4, 16, 4), (8, 4, 4, 32, 4), (33, 8, 2, 32, 16),
        (33, 4, 4, 64, 8), (8, 8, 4, 32, 4), (70, 8, 8, 64, 16),
    ]
    arrays, meta = {}, []
    g = torch.Generator().manual_seed(4321)
    for i, (t, hq, hkv, d, bs) in enumerate(cases):
        need = (t + bs - 1) // bs
        nb = need + 1
        cache = torch.randn(2, nb, hkv, bs, d, generator=g).half().float()
        q = torch.randn(t, hq, d, generator=g).half().float()
        k = torch.randn(t, hkv, d, generator=g).half().float()
        v = torch.randn(t, hkv, d, generator=g).half().float()
        bt = torch.randperm(nb, generator=g)[:need].tolist()
        arrays[f'c{i}_cache_before'] = cache.numpy().astype(np.float16)
        out = ref_pa.paged_attention_prefill(q, k, v, cache, bt, bs)
        arrays[f'c{i}_query'] = q.numpy().astype(np.float16)
        arrays[f'c{i}_key'] = k.numpy().astype(np.float16)
        arrays[f'c{i}_value'] = v.numpy().astype(np.float16)
        arrays[f'c{i}_block_table'] = np.array(bt)
        arrays[f'c{i}_out'] = out.contiguous().numpy()
        arrays[f'c{i}_cache_after'] = cache.numpy().astype(np.float16)
        meta.append(dict(t=t, hq=hq, hkv=hkv, d=d, bs=bs, nb=nb))
    arrays['meta'] = np.array(json.dumps(meta))
    save_npz('g3_prefill_cases.npz', **arrays)


# ---------------------------------------------------------------- G4
def g4():
    rng = np.random.RandomState(7)
    cases = [dict(block_table=[3, 7], seq_len=6, block_size=4)]
    for bs in (1, 4, 8, 16, 3):
        for _ in range(3):
            nblk = int(rng.randint(1, 9))
            bt = rng.permutation(64)[:nblk].tolist()
            seq_len = int(rng.randint(0, nblk * bs + 1))
            cases.append(dict(block_table=bt, seq_len=seq_len, block_size=bs))
    for c in cases:
        c['slots'] = ref_pa.compute_slot_mapping(c['block_table'], c['seq_len'], c['block_size'])
    save_json('g4_slot_mapping.json', cases)


# ---------------------------------------------------------------- G5
def g5():
    # KVCachePool (01): scripted alloc/free
    pool = ref_pa.KVCachePool(num_blocks=8, block_size=4, num_kv_heads=1, head_dim=4, num_layers=1)
    script = [['alloc', 3], ['alloc', 2], ['free', [1]], ['free', [0, 4]], ['alloc', 4],
              ['alloc', 99], ['free', [2, 3]], ['alloc', 1], ['alloc', 5], ['alloc', 1]]
    pool_trace = []
    held = []
    for op, arg in script:
        rec = dict(op=op, arg=arg)
        if op == 'alloc':
            try:
                rec['result'] = pool.allocate(arg)
            except RuntimeError as e:
                rec['error'] = str(e)
        else:
            pool.free(arg)
        rec['free_blocks'] = list(pool.free_blocks)
        rec['ref_counts'] = list(pool.ref_counts)
        pool_trace.append(rec)
    # BlockAllocator (05): LRU order, prefix-hash reuse, OOM -> None
    alloc = ref_mv.BlockAllocator(num_blocks=6, block_size=16)
    script2 = [['alloc', None], ['alloc', None], ['alloc', None], ['mark', 1, 42], ['free', 1],
               ['free', 0], ['alloc', 42], ['alloc', None], ['alloc', None], ['alloc', None],
               ['alloc', None], ['alloc', None], ['free', 1], ['free', 1], ['free', 1],
               ['alloc', 42], ['mark', 2, 7], ['mark', 3, 7], ['free', 2], ['alloc', None],
               ['alloc', 7], ['free', 5], ['free', 4], ['alloc', None], ['alloc', None]]
    alloc_trace = []
    for item in script2:
        op = item[0]
        rec = dict(op=op, arg=item[1:])
        if op == 'alloc':
            rec['result'] = alloc.allocate(item[1]) if item[1] is not None else alloc.allocate()
        elif op == 'free':
            alloc.free(item[1])
        else:
            alloc.mark_cached(item[1], item[2])
        rec['free_order'] = list(alloc.free_blocks.keys())
        rec['ref_count'] = list(alloc.ref_count)
        rec['num_free'] = alloc.num_free
        rec['hash_to_block'] = {str(k): v for k, v in alloc.hash_to_block.items()}
        alloc_trace.append(rec)
    save_json('g5_allocator_traces.json', dict(
        pool=dict(num_blocks=8, trace=pool_trace),
        allocator=dict(num_blocks=6, block_size=16, trace=alloc_trace)))


# ---------------------------------------------------------------- G6
def _drive(cfg, requests, token_rule, max_steps=200):
    alloc = ref_mv.BlockAllocator(num_blocks=cfg['num_blocks'], block_size=cfg['block_size'])
    sched = ref_mv.Scheduler(max_num_seqs=cfg['max_num_seqs'],
                             max_num_batched_tokens=cfg['max_num_batched_tokens'],
                             block_size=cfg['block_size'], allocator=alloc,
                             enable_chunked_prefill=cfg['chunked'])
    for r in requests:
        sched.add_request(ref_mv.Request(r['id'], list(r['prompt']), r['max_new_tokens']))
    steps = []
    step = 0
    while sched.has_unfinished and step < max_steps:
        batch = sched.schedule()
        if not batch:
            steps.append(dict(scheduled=[], next_tokens=[], num_free=alloc.num_free,
                              waiting=[r.request_id for r in sched.waiting],
                              running=[r.request_id for r in sched.running]))
            break
        rec = [dict(id=s.request.request_id, tokens=list(s.tokens), positions=list(s.positions),
                    is_prefill=bool(s.is_prefill), slot_mapping=list(s.slot_mapping),
                    block_table=list(s.request.block_table)) for s in batch]
        nxt = [token_rule(step, i) for i in range(len(batch))]
        sched.update(batch, nxt)
        steps.append(dict(scheduled=rec, next_tokens=nxt, num_free=alloc.num_free,
                          waiting=[r.request_id for r in sched.waiting],
                          running=[r.request_id for r in sched.running],
                          finished=[r.request_id for r in sched.finished]))
        step += 1
    outputs = {r.request_id: list(r.output_token_ids) for r in sched.finished}
    return dict(config=cfg, requests=requests, steps=steps, outputs=outputs)


def g6():
    def rule_plain(step, i):
        return (step * 7 + i * 3) % 50 + 1

    def rule_eos(step, i):
        return 0 if (step == 3 and i == 0) else (step * 5 + i) % 30 + 1

    scen = {}
    base = dict(num_blocks=64, block_size=16, max_num_seqs=8)
    scen['chunked_budget32'] = _drive(dict(base, max_num_batched_tokens=32, chunked=True),
                                      [dict(id='r0', prompt=list(range(64)), max_new_tokens=5)], rule_plain)
    scen['chunked_budget16'] = _drive(dict(base, max_num_batched_tokens=16, chunked=True),
                                      [dict(id='r0', prompt=list(range(64)), max_new_tokens=4)], rule_plain)
    scen['unchunked_over_budget'] = _drive(dict(base, max_num_batched_tokens=64, chunked=False),
                                           [dict(id='r0', prompt=list(range(1, 129)), max_new_tokens=3),
                                            dict(id='r1', prompt=[5, 6, 7], max_new_tokens=3)], rule_plain)
    scen['max_num_seqs2'] = _drive(dict(num_blocks=128, block_size=16, max_num_seqs=2,
                                        max_num_batched_tokens=256, chunked=True),
                                   [dict(id=f'r{i}', prompt=[i + 1, i + 2, i + 3], max_new_tokens=3 + i)
                                    for i in range(5)], rule_plain)
    scen['oom_mid_decode'] = _drive(dict(num_blocks=3, block_size=4, max_num_seqs=4,
                                         max_num_batched_tokens=64, chunked=True),
                                    [dict(id='r0', prompt=[1, 2, 3, 4], max_new_tokens=6),
                                     dict(id='r1', prompt=[5, 6, 7], max_new_tokens=8),
                                     dict(id='r2', prompt=[9, 9, 9, 9, 9], max_new_tokens=2)], rule_plain,
                                    max_steps=40)
    scen['eos_finish'] = _drive(dict(base, max_num_batched_tokens=64, chunked=True),
                                [dict(id='r0', prompt=[3, 1, 4, 1, 5], max_new_tokens=10),
                                 dict(id='r1', prompt=[2, 7], max_new_tokens=6)], rule_eos)
    scen['mixed_chunked_decode'] = _drive(dict(num_blocks=64, block_size=8, max_num_seqs=4,
                                               max_num_batched_tokens=24, chunked=True),
                                          [dict(id='r0', prompt=list(range(1, 41)), max_new_tokens=4),
                                           dict(id='r1', prompt=[8, 9], max_new_tokens=6),
                                           dict(id='r2', prompt=list(range(50, 63)), max_new_tokens=3)],
                                          rule_plain)
    save_json('g6_scheduler_traces.json', scen)


# ---------------------------------------------------------------- G7
def g7():
    nan = float('nan')
    rows = [[1.0, 5.0, 2.0], [3.0, 1.0, 4.0], [1.0, 5.0, 5.0], [7.0, 7.0, 7.0], [1.0, nan, 7.0],
            [nan, 2.0, nan], [-1.0, -3.0, -2.0], [float('-inf'), -5.0, float('-inf')],
            [0.0, float('inf'), float('inf')]]
    greedy = []
    for row in rows:
        req = ref_mv.Request('r', [1], max_new_tokens=1, temperature=0.0)
        greedy.append(ref_mv.Sampler.sample(torch.tensor([row]), [req])[0])
    g = torch.Generator().manual_seed(99)
    big = torch.randn(6, 257, generator=g)
    big[2, 100] = big[2].max() + 0.0   # a tie for the row maximum
    big[2, 30] = big[2].max()
    greedy_big = ref_mv.Sampler.sample(
        big, [ref_mv.Request('r', [1], 1, temperature=0.0) for _ in range(big.shape[0])])

    # nucleus intermediates: capture what the reference hands to torch.multinomial
    # (the renormalised, sorted, masked distribution) and recover the token order by
    # forcing the drawn index.
    cases = []
    logits = torch.randn(5, 50, generator=g) * 2.0
    params = [(1.0, 0.9), (0.7, 0.5), (1.5, 0.95), (0.3, 0.8), (1.0, 0.2)]
    real_multinomial = torch.multinomial
    try:
        for row, (temp, top_p) in zip(logits, params):
            seen = {}

            def fake(probs, n, forced=[0]):
                seen['probs'] = probs.clone()
                return torch.tensor([forced[0]])

            req = ref_mv.Request('r', [1], 1, temperature=temp, top_p=top_p)
            torch.multinomial = lambda p, n: fake(p, n, [0])
            ref_mv.Sampler.sample(row[None, :], [req])
            sorted_probs = seen['probs']
            kept = int((sorted_probs > 0).sum())
            order = []
            for j in range(kept):
                torch.multinomial = lambda p, n, j=j: fake(p, n, [j])
                order.append(ref_mv.Sampler.sample(row[None, :], [req])[0])
            cases.append(dict(temperature=temp, top_p=top_p, logits=row.tolist(),
                              kept_ids_sorted=order,
                              kept_probs_sorted=sorted_probs[:kept].tolist()))
    finally:
        torch.multinomial = real_multinomial
    save_json('g7_sampler.json', dict(greedy_rows=[[None if (isinstance(x, float) and math.isnan(x)) else
                                                     ('inf' if x == float('inf') else '-inf' if x == float('-inf') else x)
                                                     for x in r] for r in rows],
                                      greedy_ids=greedy, nucleus=cases))
    save_npz('g7_sampler_big.npz', logits=big.numpy(), greedy_ids=np.array(greedy_big))


def g7b():
    """Nucleus rows at the benchmark's vocabulary (32000): what the reference's Sampler keeps.
    Captured from inside Sampler.sample by observing its Tensor.sort (token order) and the
    distribution it hands to torch.multinomial (renormalised, cut entries zero).  `margin` is
    how far the exclusive cumulative mass at the cut is from top_p on either side, in units of
    the row's largest fp32 cumulative-sum spacing (2^-24): a kept set is only well defined
    against another implementation's rounding when that is comfortably above 1."""
    g = torch.Generator().manual_seed(32000)
    V = 32000
    params = [(1.0, 0.9), (0.7, 0.5), (1.3, 0.95), (0.5, 0.99)]
    logits = torch.randn(len(params), V, generator=g) * torch.tensor([[1.0], [2.0], [1.5], [3.0]])
    arrays = dict(logits=logits.numpy(), temperature=np.array([t for t, _ in params], dtype=np.float32),
                  top_p=np.array([p for _, p in params], dtype=np.float32))
    real_sort, real_multinomial = torch.Tensor.sort, torch.multinomial
    try:
        for i, (row, (temp, top_p)) in enumerate(zip(logits, params)):
            seen = {}

            def spy_sort(*a, **kw):
                out = real_sort(*a, **kw)
                seen['ids'] = out[1].clone()
                seen['sorted'] = out[0].clone()
                return out

            def fake_multinomial(probs, n):
                seen['probs'] = probs.clone()
                return torch.tensor([0])

            torch.Tensor.sort, torch.multinomial = spy_sort, fake_multinomial
            ref_mv.Sampler.sample(row[None, :], [ref_mv.Request('r', [1], 1, temperature=temp, top_p=top_p)])
            kept = int((seen['probs'] > 0).sum())
            excl = torch.cumsum(seen['sorted'], dim=-1) - seen['sorted']
            margin = min(float(top_p - excl[kept - 1]), float(excl[kept] - top_p)) / 2.0 ** -24
            arrays[f'kept_ids_sorted_{i}'] = seen['ids'][:kept].numpy().astype(np.int32)
            arrays[f'kept_probs_sorted_{i}'] = seen['probs'][:kept].numpy()
            arrays[f'margin_{i}'] = np.array(margin)
            print(f'  g7b row {i}: T={temp} top_p={top_p}: kept {kept} of {V}, margin {margin:.1f} x 2^-24')
    finally:
        torch.Tensor.sort, torch.multinomial = real_sort, real_multinomial
    save_npz('g7_sampler_vocab32000.npz', **arrays)


# ---------------------------------------------------------------- G8
def g8():
    torch.manual_seed(2024)
    cfg = dict(vocab_size=100, hidden_size=64, num_layers=2, num_heads=4, intermediate_size=128,
               num_kv_blocks=64, block_size=8, max_num_seqs=4, max_num_batched_tokens=128)
    eng = ref_mv.MiniVLLM(**cfg)
    params = {k: v.detach().numpy().copy() for k, v in eng.model.state_dict().items()
              if not (k.endswith('_cached') or k.endswith('inv_freq') or k in ('k_cache', 'v_cache'))}
    prompts = [[1, 2, 3, 4, 5], [10, 20, 30], [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17]]
    for p in prompts:
        eng.add_request(p, max_new_tokens=4, temperature=0.0)
    logits_log = []
    real_forward = eng.model.forward

    def spy(*a, **kw):
        out = real_forward(*a, **kw)
        logits_log.append(out.detach().numpy().copy())
        return out

    eng.model.forward = spy
    step_results = []
    while eng.scheduler.has_unfinished and len(step_results) < 20:
        step_results.append(eng.step())
    arrays = {f'param::{k}': v for k, v in params.items()}
    for i, lg in enumerate(logits_log):
        arrays[f'logits_step{i}'] = lg
    arrays['meta'] = np.array(json.dumps(dict(config=cfg, prompts=prompts, max_new_tokens=4,
                                              step_results=step_results)))
    save_npz('g8_engine_greedy.npz', **arrays)




# ---------------------------------------------------------------- G9 (MLA, 04_mla/mini_mla.py)
def g9():
    sys.path.insert(0, os.path.join(REF, '04_mla'))
    import mini_mla as ref_mla
    arrays, meta = {}, []
    cfgs = [dict(hidden_size=128, num_heads=4, qk_nope_head_dim=16, qk_rope_head_dim=8, kv_lora_rank=32,
                 v_head_dim=16, max_seq_len=256),          # the reference test-suite's small_mla
            dict(hidden_size=128, num_heads=8, qk_nope_head_dim=32, qk_rope_head_dim=16, kv_lora_rank=64,
                 v_head_dim=32, max_seq_len=256)]
    for ci, cfg in enumerate(cfgs):
        torch.manual_seed(100 + ci)
        m = ref_mla.MultiHeadLatentAttention(**cfg).eval()
        with torch.no_grad():
            for prm in m.parameters():                      # fp16-representable weights: half the fixture size
                prm.copy_(prm.half().float())
        for k, v in m.state_dict().items():
            if 'rotary' not in k:
                arrays[f'm{ci}_w::{k}'] = v.numpy().astype(np.float16)
        g = torch.Generator().manual_seed(7 + ci)
        B, T0 = 2, 12 + ci * 9
        calls = []
        with torch.no_grad():
            x = torch.randn(B, T0, cfg['hidden_size'], generator=g)
            pos = torch.arange(T0).unsqueeze(0).expand(B, -1)
            out, c_kv, k_rope = m(x, pos)
            calls.append(('prefill', x, pos, None, None, out, c_kv, k_rope))
            for step in range(3):
                xd = torch.randn(B, 1, cfg['hidden_size'], generator=g)
                pd = torch.full((B, 1), c_kv.shape[1])
                out, nc, nr = m(xd, pd, past_c_kv=c_kv, past_k_rope=k_rope)
                calls.append((f'decode{step}', xd, pd, c_kv, k_rope, out, nc, nr))
                c_kv, k_rope = torch.cat([c_kv, nc], 1), torch.cat([k_rope, nr], 1)
            xc = torch.randn(B, 5, cfg['hidden_size'], generator=g)
            pc = torch.arange(c_kv.shape[1], c_kv.shape[1] + 5).unsqueeze(0).expand(B, -1)
            out, nc, nr = m(xc, pc, past_c_kv=c_kv, past_k_rope=k_rope)
            calls.append(('chunk', xc, pc, c_kv, k_rope, out, nc, nr))
        names = []
        for name, x, pos, pc_, pr_, out, nc, nr in calls:
            names.append(name)
            pre = f'm{ci}_{name}_'
            arrays[pre + 'x'] = x.numpy()
            arrays[pre + 'pos'] = pos.numpy()
            if pc_ is not None:
                arrays[pre + 'past_c_kv'] = pc_.numpy()
                arrays[pre + 'past_k_rope'] = pr_.numpy()
            arrays[pre + 'out'] = out.numpy()
            arrays[pre + 'new_c_kv'] = nc.numpy()
            arrays[pre + 'new_k_rope'] = nr.numpy()
        meta.append(dict(config=cfg, calls=names))
    mha, mla = ref_mla.compare_kv_cache_size(num_heads=32, head_dim=128, kv_lora_rank=512, qk_rope_head_dim=64,
                                             num_layers=32, seq_len=4096, batch_size=1)
    arrays['meta'] = np.array(json.dumps(dict(models=meta, kv_size_mb=[mha, mla])))
    save_npz('g9_mla.npz', **arrays)


if __name__ == '__main__':
    only = [a for a in sys.argv[2:]]           # e.g. `make_golden.py /root/reference g7b`: just those fixtures
    for fn in (g1, g2, g3, g4, g5, g6, g7, g7b, g8, g9):
        if only and fn.__name__ not in only:
            continue
        fn()
        print('wrote', fn.__name__)
