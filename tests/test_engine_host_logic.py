"""Host-side planning of the engine (CPU only): which kernel serves which token of a flattened
step, and the persistent device block-table mirror (exercised here on torch's CPU device)."""
import pytest
import torch

from mini_vllm import DeviceBlockTables, Request, build_attn_plan


def _req(rid, table):
    r = Request(rid, [1], 1)
    r.block_table = list(table)
    return r


def _rows(index):
    """Token rows a plan entry stands for: a slice when they are consecutive, else a tensor."""
    return list(range(index.start, index.stop)) if isinstance(index, slice) else index.tolist()


def test_plan_all_prefill_from_position_zero_uses_one_causal_segment_per_request():
    plan = build_attn_plan([0, 1, 2, 0, 1], [3, 2], [True, True], [[4], [9]], 'cpu')
    assert plan.total_tokens == 5 and plan.causal_tokens == 5 and plan.paged_rows == 0
    assert plan.causal_index is None                      # every token -> no gather needed
    assert plan.cu_seqlens.tolist() == [0, 3, 5] and plan.max_causal_len == 3


def test_plan_mixed_step_true_context():
    # request 0: decode at position 7; request 1: later chunk (positions 4..6); request 2: first chunk
    positions = [7, 4, 5, 6, 0, 1]
    plan = build_attn_plan(positions, [1, 3, 2], [False, True, True], [[3, 8], [5], [2]], 'cpu')
    assert plan.total_tokens == 6
    assert _rows(plan.causal_index) == [4, 5] and plan.cu_seqlens.tolist() == [0, 2]
    assert _rows(plan.paged_index) == [0, 1, 2, 3]
    assert plan.paged_context.tolist() == [8, 5, 6, 7] and plan.max_context == 8
    assert plan.paged_tables.tolist() == [[3, 8], [5, 0], [5, 0], [5, 0]]      # padded with 0


def test_plan_reference_semantics_without_block_tables():
    """No block tables (or reference_compat): prompt chunks attend within the chunk only and a
    decode row reads flat slot 0, like 05_mini_vllm/mini_vllm.py:437-481."""
    for kwargs in (dict(block_tables=None), dict(block_tables=[[3], [4]], reference_compat=True)):
        plan = build_attn_plan([5, 9, 10], [1, 2], [False, True], kwargs.pop('block_tables'), 'cpu', **kwargs)
        assert _rows(plan.causal_index) == [1, 2] and plan.cu_seqlens.tolist() == [0, 2]
        assert _rows(plan.paged_index) == [0]
        assert plan.paged_tables.tolist() == [[0]] and plan.paged_context.tolist() == [1]


def test_device_block_tables_incremental_sync_and_reuse():
    tabs = DeviceBlockTables('cpu', rows=2, width=2)
    a, b, c = _req('a', [5]), _req('b', [7, 8, 9]), _req('c', [1])
    rows = tabs.sync([a, b, c])                            # needs more rows AND more width
    assert len(set(rows)) == 3
    got = tabs.gather(rows)
    assert got[0, :1].tolist() == [5] and got[1, :3].tolist() == [7, 8, 9] and got[2, :1].tolist() == [1]
    a.block_table.append(6)                                # tables only grow at the end
    b.block_table += [10, 11]
    assert tabs.sync([a, b, c]) == rows                    # rows are stable while a request runs
    got = tabs.gather(rows)
    assert got[0, :2].tolist() == [5, 6] and got[1, :5].tolist() == [7, 8, 9, 10, 11]
    tabs.release('b')
    d = _req('d', [42])
    new_rows = tabs.sync([a, d])
    assert new_rows[0] == rows[0] and new_rows[1] == rows[1]          # freed row is reused
    assert tabs.gather(new_rows)[1, :1].tolist() == [42]
    plan = build_attn_plan([8, 3], [1, 1], [False, False], None, 'cpu', device_tables=tabs, table_rows=new_rows)
    assert plan.paged_rows == 2 and plan.paged_context.tolist() == [9, 4]
    assert plan.paged_tables[0, :2].tolist() == [5, 6] and plan.paged_tables[1, 0].item() == 42


def test_device_block_tables_many_requests_growth():
    tabs = DeviceBlockTables('cpu', rows=2, width=1)
    reqs = [_req(f'r{i}', list(range(i, i + 1 + i % 5))) for i in range(37)]
    rows = tabs.sync(reqs)
    assert sorted(rows) == sorted(set(rows))
    got = tabs.gather(rows)
    for i, r in enumerate(reqs):
        assert got[i, :len(r.block_table)].tolist() == r.block_table


# ------------------------------------------------------------------ decode-only scheduler path
def _snapshot(sched):
    alloc = sched.allocator
    return dict(running=[(r.request_id, r.num_computed_tokens, list(r.output_token_ids), list(r.block_table),
                          r.is_finished) for r in sched.running],
                finished=[r.request_id for r in sched.finished],
                free=list(alloc.free_blocks.items()), refs=list(alloc.ref_count))


def _drive(fast, num_blocks, nreq, prompt_lens, max_new, eos_every=0, chunked=True):
    """Run a scheduler to completion feeding deterministic 'sampled' tokens; with fast=True every
    step first tries schedule_decode_only().  Returns the per-step trace and how often the
    decode-only path served a step."""
    from mini_vllm import BlockAllocator, Scheduler
    sched = Scheduler(max_num_seqs=6, max_num_batched_tokens=24, block_size=4,
                      allocator=BlockAllocator(num_blocks, 4), enable_chunked_prefill=chunked)
    for i in range(nreq):
        sched.add_request(Request(f"r{i}", list(range(1, prompt_lens[i % len(prompt_lens)] + 1)), max_new[i % len(max_new)]))
    trace, served, step = [], 0, 0
    while sched.has_unfinished and step < 400:
        step += 1
        batch = sched.schedule_decode_only() if fast else None
        if batch is not None:
            served += 1
            feed = [(r.request_id, [r.output_token_ids[-1]], [r.num_computed_tokens], False,
                     [r.block_table[r.num_computed_tokens // 4] * 4 + r.num_computed_tokens % 4]) for r in batch]
            toks = [0 if eos_every and (step + j) % eos_every == 0 else 7 + (step + j) % 5 for j in range(len(batch))]
            sched.update_decode_only(batch, toks)
        else:
            items = sched.schedule()
            feed = [(it.request.request_id, list(it.tokens), list(it.positions), it.is_prefill, list(it.slot_mapping))
                    for it in items]
            toks = [0 if eos_every and (step + j) % eos_every == 0 else 7 + (step + j) % 5 for j in range(len(items))]
            sched.update(items, toks)
        trace.append((feed, _snapshot(sched)))
    assert not sched.has_unfinished
    return trace, served


def test_decode_only_path_reproduces_the_general_scheduler():
    """schedule_decode_only / update_decode_only leave requests, block tables and the allocator
    (free-list order, refcounts) exactly as schedule / update do, step for step -- including runs
    where blocks run short (the path declines and the general scheduler skips requests), requests
    finish by EOS mid-batch, and requests wait for a sequence slot behind running decodes."""
    cases = [dict(num_blocks=64, nreq=5, prompt_lens=[3, 9, 4], max_new=[9, 5, 14]),
             dict(num_blocks=9, nreq=4, prompt_lens=[7, 5], max_new=[3, 12]),          # temporary block shortage
             dict(num_blocks=64, nreq=9, prompt_lens=[2, 11, 6], max_new=[8, 3], eos_every=7),
             dict(num_blocks=40, nreq=8, prompt_lens=[30, 3], max_new=[6], chunked=False),
             # more requests than sequence slots: the path also serves steps in which requests wait
             # for a slot (nobody can be admitted), and must decline the moment one frees
             dict(num_blocks=64, nreq=14, prompt_lens=[5, 2, 9], max_new=[4, 11, 7], eos_every=5)]
    for kw in cases:
        slow, none_served = _drive(False, **kw)
        fast, served = _drive(True, **kw)
        assert none_served == 0 and served > 0
        assert len(slow) == len(fast)
        for a, b in zip(slow, fast):
            assert a == b


def test_decode_only_path_declines_without_side_effects():
    from mini_vllm import BlockAllocator, Scheduler
    sched = Scheduler(4, 16, 4, BlockAllocator(3, 4))
    assert sched.schedule_decode_only() is None                       # nothing running
    a = Request("a", [1, 2, 3, 4], 8)
    b = Request("b", [1, 2, 3, 4], 8)
    sched.add_request(a)
    sched.add_request(b)
    assert sched.schedule_decode_only() is None                       # requests still waiting
    items = sched.schedule()
    sched.update(items, [5, 5])
    assert [r.request_id for r in sched.running] == ["a", "b"] and sched.allocator.num_free == 1
    before = _snapshot(sched)
    assert sched.schedule_decode_only() is None                       # both need a block, one is free
    assert _snapshot(sched) == before
    sched.max_num_seqs = 1
    assert sched.schedule_decode_only() is None                       # more running than the seq cap


# ------------------------------------------------------------------ prefix caching (host logic)
def _prefix_sched(num_blocks=32, bs=4, budget=64, **kw):
    from mini_vllm import BlockAllocator, Scheduler
    return Scheduler(max_num_seqs=8, max_num_batched_tokens=budget, block_size=bs,
                     allocator=BlockAllocator(num_blocks, bs), enable_prefix_caching=True, **kw)


def test_prompt_block_hashes_chain():
    from mini_vllm import prompt_block_hashes
    a = prompt_block_hashes([1, 2, 3, 4, 5, 6, 7, 8, 9], 4)
    assert len(a) == 2                                              # only full blocks
    assert a == prompt_block_hashes([1, 2, 3, 4, 5, 6, 7, 8], 4)
    b = prompt_block_hashes([9, 2, 3, 4, 5, 6, 7, 8], 4)
    assert a[0] != b[0] and a[1] != b[1]                            # same 2nd block, different prefix
    assert a[0] == hash((0, (1, 2, 3, 4), None))                    # the reference's formula, seed 0
    assert prompt_block_hashes([1, 2, 3], 4) == []


def test_prefix_cache_shares_prompt_blocks_between_requests():
    sched = _prefix_sched()
    shared = list(range(1, 13))                                     # 3 full blocks
    a = Request("a", shared + [50, 51], 4)
    sched.add_request(a)
    items = sched.schedule()
    assert items[0].positions[0] == 0 and len(items[0].tokens) == 14 and a.num_cached_tokens == 0
    sched.update(items, [7])
    assert a.num_registered_blocks == 3
    assert [sched.allocator.block_to_hash[b] for b in a.block_table[:3]] == a.block_hashes
    # a second request with the same 12-token prefix adopts those three blocks
    b = Request("b", shared + [60, 61, 62], 4)
    sched.add_request(b)
    items = sched.schedule()
    by_id = {it.request.request_id: it for it in items}
    assert b.block_table[:3] == a.block_table[:3] and b.num_cached_tokens == 12
    assert by_id["b"].tokens == [60, 61, 62] and by_id["b"].positions == [12, 13, 14] and by_id["b"].is_prefill
    assert by_id["b"].slot_mapping == [b.block_table[3] * 4 + i for i in range(3)]
    assert all(sched.allocator.ref_count[blk] == 2 for blk in a.block_table[:3])
    assert b.block_table[3] not in a.block_table                    # the tail is private
    assert sched.prefix_stats == {'requests': 2, 'prompt_tokens': 29, 'cached_tokens': 12}
    # finishing one owner keeps the blocks alive for the other
    sched.update(items, [0 if it.request is a else 9 for it in items])       # a hits EOS
    assert a.is_finished and all(sched.allocator.ref_count[blk] == 1 for blk in b.block_table[:3])


def test_fully_cached_prompt_still_computes_its_last_block():
    sched = _prefix_sched()
    prompt = list(range(1, 9))                                      # exactly 2 full blocks
    first = Request("a", list(prompt), 1)
    sched.add_request(first)
    sched.update(sched.schedule(), [5])
    assert first.is_finished and sched.allocator.num_free == 32     # blocks are free but keep their hashes
    again = Request("b", list(prompt), 1)
    sched.add_request(again)
    items = sched.schedule()
    assert again.num_cached_tokens == 4 and items[0].positions == [4, 5, 6, 7]
    assert again.block_table[0] == first.block_table[0]             # revived from the free list
    assert sched.allocator.ref_count[again.block_table[0]] == 1
    assert again.block_table[0] not in sched.allocator.free_blocks


def test_evicted_blocks_stop_matching():
    sched = _prefix_sched(num_blocks=4)
    a = Request("a", list(range(1, 9)) + [9], 1)                    # 3 blocks
    sched.add_request(a)
    sched.update(sched.schedule(), [5])
    assert a.is_finished and len(sched.allocator.hash_to_block) == 2
    filler = Request("f", list(range(100, 116)), 1)                 # needs all 4 blocks -> evicts both cached ones
    sched.add_request(filler)
    sched.update(sched.schedule(), [5])
    assert not any(h in sched.allocator.hash_to_block for h in a.block_hashes)
    b = Request("b", list(range(1, 9)) + [9], 1)
    sched.add_request(b)
    items = sched.schedule()
    assert b.num_cached_tokens == 0 and items[0].positions[0] == 0


def test_prefix_caching_off_changes_nothing():
    """The default scheduler never touches the hash map (as in the reference)."""
    from mini_vllm import BlockAllocator, Scheduler
    sched = Scheduler(4, 64, 4, BlockAllocator(16, 4))
    for name in ("a", "b"):
        sched.add_request(Request(name, list(range(1, 10)), 2))
    for _ in range(3):
        items = sched.schedule()
        sched.update(items, [3] * len(items))
    assert not sched.allocator.hash_to_block and sched.prefix_stats['requests'] == 0


def test_chunked_prefill_registers_blocks_as_chunks_complete():
    sched = _prefix_sched(budget=6)
    a = Request("a", list(range(1, 15)), 2)                         # 14 tokens: chunks of 6, 6, 2
    sched.add_request(a)
    seen = []
    while a.is_prefill:
        items = sched.schedule()
        sched.update(items, [4])
        seen.append((a.num_computed_tokens, a.num_registered_blocks))
    assert seen == [(6, 1), (12, 3), (14, 3)]


def test_plan_later_chunks_become_append_segments_when_the_kernel_covers_the_model():
    # request 0: decode at position 7; request 1: later chunk (positions 4..6); request 2: first chunk;
    # request 3: a one-token later chunk (stays a paged-decode row)
    positions = [7, 4, 5, 6, 0, 1, 9]
    args = (positions, [1, 3, 2, 1], [False, True, True, True], [[3, 8], [5, 6], [2], [4, 1, 7]], 'cpu')
    plan = build_attn_plan(*args, append_ok=True)
    assert _rows(plan.causal_index) == [4, 5] and plan.cu_seqlens.tolist() == [0, 2]
    assert _rows(plan.paged_index) == [0, 6] and plan.paged_context.tolist() == [8, 10]
    assert isinstance(plan.causal_index, slice) and not isinstance(plan.paged_index, slice)   # views where rows are consecutive
    assert plan.paged_tables.tolist() == [[3, 8, 0], [4, 1, 7]]
    assert plan.append_tokens == 3 and plan.append_meta.tolist() == [[1], [3], [4]]
    assert plan.append_tables.tolist() == [[5, 6]]
    assert plan.max_append_len == 3 and plan.max_append_context == 7
    # without the kernel the same tokens are paged-decode rows
    rows = build_attn_plan(*args)
    assert rows.append_tokens == 0 and _rows(rows.paged_index) == [0, 1, 2, 3, 6]
    assert rows.paged_context.tolist() == [8, 5, 6, 7, 10]
    # a per-segment rule: contexts the table-staging kernels cannot hold fall back to rows
    limited = build_attn_plan(*args, append_ok=lambda n_q, context: context <= 6)
    assert limited.append_tokens == 0 and _rows(limited.paged_index) == [0, 1, 2, 3, 6]
    assert build_attn_plan(*args, append_ok=lambda n_q, context: context <= 7).append_tokens == 3


def test_paged_prefill_serves_rule():
    """The rule is the library's own (mi355_paged_prefill_max_context: host arithmetic, no device)."""
    import torch
    from mi355_paged import ops
    limit = (8190 * 16 // 1024) * 1024     # blocks one workgroup stages, in tokens, down to the kernels' token quantum
    assert ops.paged_prefill_serves(torch.float32, 16, 16, 8, limit)
    assert not ops.paged_prefill_serves(torch.float32, 16, 16, 8, limit + 1)        # VALU kernel stages the table
    assert not ops.paged_prefill_serves(torch.bfloat16, 64, 16, 47, limit + 1)      # short chunk: append kernel, staged
    assert ops.paged_prefill_serves(torch.bfloat16, 64, 16, 48, 10 * limit)         # flash kernel reads the table itself
    assert ops.paged_prefill_serves(torch.float32, 64, 16, 48, 10 * limit)          # ... and so does the fp32 one
    assert not ops.paged_prefill_serves(torch.float32, 256, 16, 48, limit + 1)      # (no fp32 flash kernel at head_dim 256)
    assert not ops.paged_prefill_serves(torch.bfloat16, 48, 16, 48, 100)            # head_dim without a kernel


# --- update() split in two halves, for an engine that issues step i + 1 before reading step i's ids

def _token_rule(req, k):
    """Token k of a request: depends on the request and k only, 0 (end of sequence) now and then."""
    return (int(req.request_id.split('-')[1]) * 7 + k * 5 + 3) % 11


def _split_sched(seed, chunked, prefix=False):
    from mini_vllm import BlockAllocator, Scheduler
    import random
    rng = random.Random(seed)
    alloc = BlockAllocator(num_blocks=96, block_size=4)
    sched = Scheduler(max_num_seqs=5, max_num_batched_tokens=24, block_size=4, allocator=alloc,
                      enable_chunked_prefill=chunked, enable_prefix_caching=prefix)
    for i in range(20):
        plen = rng.randint(1, 22)
        prompt = [1 + (i % 3)] * plen if prefix else [rng.randint(1, 50) for _ in range(plen)]
        sched.add_request(Request(request_id=f'req-{i}', prompt_token_ids=prompt, max_new_tokens=rng.randint(1, 9)))
    return sched, alloc


def _trace_item(item):
    return (item.request.request_id, item.start, len(item.tokens), item.is_prefill, tuple(item.request.block_table))


@pytest.mark.parametrize('chunked', [True, False])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_advance_then_resolve_equals_update(seed, chunked):
    a, alloc_a = _split_sched(seed, chunked)
    b, alloc_b = _split_sched(seed, chunked)
    steps = 0
    while a.has_unfinished:
        sa, sb = a.schedule(), b.schedule()
        assert [_trace_item(x) for x in sa] == [_trace_item(x) for x in sb]
        if not sa:
            break
        toks = [_token_rule(x.request, len(x.request.output_token_ids)) for x in sa]
        a.update(sa, toks)
        takes = b.advance(sb)
        assert all(x.request.output_token_ids[-1] == -1 - j for j, x in enumerate(sb) if takes[j])
        b.resolve(sb, takes, toks)
        assert alloc_a.num_free == alloc_b.num_free
        assert [r.request_id for r in a.running] == [r.request_id for r in b.running]
        steps += 1
    assert steps > 10 and not b.has_unfinished
    assert ({r.request_id: r.output_token_ids for r in a.finished}
            == {r.request_id: r.output_token_ids for r in b.finished})
    assert any(r.output_token_ids[-1] == 0 for r in a.finished)          # ends on the end-of-sequence id happened
    assert any(r.output_token_ids[-1] != 0 for r in a.finished)          # and so did ends by length


@pytest.mark.parametrize('prefix', [False, True])
@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_scheduling_one_step_ahead_gives_every_request_the_same_tokens(seed, prefix):
    """The engine's pipelined loop on the scheduler alone: step i + 1 is scheduled between advance()
    and resolve() of step i.  A request that ended on the end-of-sequence id in step i has a wasted
    item in step i + 1; everybody's tokens equal those of the step-by-step loop."""
    a, alloc_a = _split_sched(seed, True, prefix)
    while a.has_unfinished:
        sa = a.schedule()
        if not sa:
            break
        a.update(sa, [_token_rule(x.request, len(x.request.output_token_ids)) for x in sa])
    want = {r.request_id: r.output_token_ids for r in a.finished}

    b, alloc_b = _split_sched(seed, True, prefix)
    cur = b.schedule()
    wasted = steps = 0
    while cur:
        wasted += sum(1 for x in cur if x.request.is_finished)
        takes = b.advance(cur)
        # the ids of this step, as the device would produce them (index = outputs before this step's stand-in)
        toks = [_token_rule(x.request, len(x.request.output_token_ids) - 1) if takes[j] else -7
                for j, x in enumerate(cur)]
        nxt = b.schedule()
        for x in nxt:                                   # decode items feed a stand-in or a known id, never garbage
            if not x.is_prefill:
                t = x.tokens[0]
                assert t >= 0 or takes[-1 - t] and cur[-1 - t].request is x.request
        b.resolve(cur, takes, toks)
        cur = nxt
        steps += 1
    assert not b.has_unfinished and steps > 10 and wasted > 0
    got = {r.request_id: r.output_token_ids for r in b.finished}
    assert got == want
    assert alloc_b.num_free == alloc_a.num_free
    assert len(b.finished) == len(set(id(r) for r in b.finished)) == 20


# --- latent (MLA) block cache: host-side bookkeeping, no device needed

def test_paged_mla_cache_blocks_rows_and_release():
    """PagedMLAKVCache on the CPU: rows of several sequences and layers land in the blocks its tables
    name (gathering them back gives what was appended, in order), blocks come from -- and return to --
    the engine's BlockAllocator, a table row is padded with block 0, and running out of blocks raises."""
    import numpy as np
    import pytest
    import torch
    import mini_mla
    from mini_vllm import BlockAllocator
    R, P, bs = 8, 4, 4
    alloc = BlockAllocator(6, bs)
    cache = mini_mla.PagedMLAKVCache(6, bs, 2, R, P, 'cpu', dtype=torch.float32, allocator=alloc)
    rng = np.random.default_rng(0)
    kept = {}

    def push(seq, layer, n):
        c = torch.from_numpy(rng.standard_normal((n, R)).astype(np.float32))
        k = torch.from_numpy(rng.standard_normal((n, P)).astype(np.float32))
        total = cache.append(seq, layer, c, k)
        old = kept.get((seq, layer))
        kept[(seq, layer)] = (c, k) if old is None else (torch.cat((old[0], c)), torch.cat((old[1], k)))
        assert total == kept[(seq, layer)][0].shape[0] == cache.length(seq, layer)

    push('a', 0, 5); push('a', 1, 5); push('b', 0, 3); push('a', 0, 2); push('a', 1, 2); push('b', 0, 1)
    assert len(cache.block_tables['a']) == 2 and len(cache.block_tables['b']) == 1 and alloc.num_free == 3
    tables = cache.tables(['b', 'a'])
    assert tables.dtype == torch.int32 and tables.shape == (2, 2) and int(tables[0, 1]) == 0
    for (seq, layer), (c, k) in kept.items():
        row = tables[['b', 'a'].index(seq)].long()
        ckv, krope = cache.views(layer)
        n = c.shape[0]
        assert torch.equal(ckv[row].reshape(-1, R)[:n], c) and torch.equal(krope[row].reshape(-1, P)[:n], k)
    with pytest.raises(RuntimeError, match='OOM'):
        push('c', 0, 4 * bs)                       # needs four blocks, three are free
    cache.free('c'); cache.free('a'); cache.free('b')
    assert alloc.num_free == 6 and not cache.block_tables and cache.length('a', 0) == 0
    with pytest.raises(ValueError):
        mini_mla.PagedMLAKVCache(4, 6, 1, R, P, 'cpu')
