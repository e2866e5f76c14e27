"""Data-parallel sharding helpers on the CPU: world_size-2 gloo processes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mi355_paged import dp


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == dp.shard_sizes(total, world)
    assert dp.shard_rows(list(range(10)), 4, 1) == [3, 4, 5]
    with pytest.raises(ValueError):
        dp.shard_bounds(4, 2, 2)


def test_all_gather_is_identity_without_process_group():
    ids = torch.tensor([5, 6, 7], dtype=torch.int32)
    assert dp.all_gather_token_ids(ids) is ids


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = torch.arange(100, 100 + total, dtype=torch.int32)      # what a single-GPU run would produce
        a, b = dp.shard_bounds(total, world, rank)
        got = dp.all_gather_token_ids(full[a:b].clone(), total=total)
        torch.save(got, os.path.join(out_dir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('total', [8, 7, 1])
def test_all_gather_world2_matches_single_process(tmp_path, total):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    want = torch.arange(100, 100 + total, dtype=torch.int32)
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f'r{r}.pt'), want)


def _overlap_worker(rank, world, port, out_dir, every=1):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n, steps = 5, 7 if every == 1 else 23
        g = dp.OverlappedTokenGather(n, 'cpu', every=every)
        seen = []
        for i in range(steps):
            buf = g.send_buffer(i)
            buf.copy_(torch.arange(n, dtype=torch.int32) + 100 * rank + 1000 * i)   # this step's "sampled ids"
            g.launch(i)
            if i >= every:
                seen.append(g.result(i - every).clone())    # an exchanged group's ids, `every` steps late
        g.flush()                                            # the last group may be partly filled
        for i in range(steps - every, steps):
            seen.append(g.result(i).clone())
        g.drain()
        torch.save(torch.stack(seen), os.path.join(out_dir, f'o{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('every', [1, 4])
def test_overlapped_gather_world2_returns_every_step_in_rank_order(tmp_path, every):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path), every), nprocs=2, join=True)
    want = torch.stack([torch.cat([torch.arange(5, dtype=torch.int32) + 100 * r + 1000 * i for r in range(2)])
                        for i in range(7 if every == 1 else 23)])
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f'o{r}.pt'), want)


def test_overlapped_gather_single_process_passthrough():
    g = dp.OverlappedTokenGather(3, 'cpu')
    g.send_buffer(0).copy_(torch.tensor([4, 5, 6], dtype=torch.int32))
    g.launch(0)
    assert g.result(0).tolist() == [4, 5, 6]
    g.drain()


def test_overlapped_gather_batched_single_process():
    g = dp.OverlappedTokenGather(2, 'cpu', every=3)
    for i in range(4):
        g.send_buffer(i).copy_(torch.tensor([10 * i, 10 * i + 1], dtype=torch.int32))
        g.launch(i)
    assert g.result(1).tolist() == [10, 11]
    with pytest.raises(RuntimeError, match='not been exchanged'):
        g.result(3)
    g.flush()
    assert g.result(3).tolist() == [30, 31]
    g.drain()


# ---------------------------------------------------------------- engine-level data parallelism
# The real scheduler, allocator and engine loop on the CPU; only the two steps that need a GPU
# (model forward, sampling) are replaced by a deterministic rule, so that the sharded run can be
# compared with a single engine on the whole batch without a device.

_ENGINE_CFG = dict(vocab_size=97, hidden_size=32, num_layers=1, num_heads=2, intermediate_size=64,
                   num_kv_blocks=64, block_size=4, max_num_seqs=3, max_num_batched_tokens=16,
                   enable_chunked_prefill=True, device='cpu',
                   reference_compat=True)      # no device-side block tables: nothing here touches a GPU
_PROMPTS = [[5, 6, 7], list(range(1, 30)), [9], [4, 4, 4, 4, 4, 4, 4, 4, 4], [11, 12], list(range(40, 61)),
            [3, 1, 4, 1, 5, 9, 2, 6]]


def _rule_engine(engine):
    """Next token = a fixed function of a request's last fed token and its position (token 0 ends
    a request early, as in the reference)."""
    vocab = engine.model.lm_head.weight.shape[0]

    def forward(token_ids, positions, slot_mapping, seq_lens, is_prefill, **kw):
        last, at = [], 0
        for n in seq_lens:
            at += n
            last.append(at - 1)
        tok, pos = token_ids[last], positions[last]
        nxt = (tok * 31 + pos * 7 + 3) % 13           # small range: token 0 (end of sequence) turns up
        return torch.nn.functional.one_hot(nxt, vocab).float()

    engine.model.forward = forward
    engine.sampler.sample = lambda logits, requests: [int(t) for t in logits.argmax(dim=-1).tolist()]
    return engine


def _dp_engine_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mini_vllm
        torch.manual_seed(100 + rank)                     # different initial weights: rank 0's are broadcast
        front = mini_vllm.DataParallelMiniVLLM(**_ENGINE_CFG)
        w = front.engine.model.lm_head.weight.detach().clone()
        _rule_engine(front.engine)
        first = front.generate(_PROMPTS, max_new_tokens=6, temperature=0.0)
        second = front.generate(_PROMPTS[:3], max_new_tokens=4, temperature=0.0)     # the front end is reusable
        torch.save(dict(first=first, second=second, lm_head=w, local=front.engine._req_counter),
                   os.path.join(out_dir, f'dp{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_engine_data_parallel_world2_equals_single_engine(tmp_path):
    import mini_vllm
    single = _rule_engine(mini_vllm.MiniVLLM(**_ENGINE_CFG))
    want = {}
    ids = [single.add_request(p, max_new_tokens=6, temperature=0.0) for p in _PROMPTS]
    want = {rid: [] for rid in ids}
    while single.scheduler.has_unfinished:
        for rid, tok in single.step():
            want[rid].append(tok)
    assert any(len(v) < 6 for v in want.values()) and any(len(v) == 6 for v in want.values())   # early ends and full runs
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_engine_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(tmp_path / f'dp{r}.pt') for r in range(2)]
    for r in range(2):
        assert got[r]['first'] == want                                   # every rank holds every request's tokens
        assert got[r]['second'] == {f'req-{7 + i}': want[f'req-{i}'][:4] for i in range(3)}
    assert torch.equal(got[0]['lm_head'], got[1]['lm_head'])             # replicated weights
    assert got[0]['local'] == 4 + 2 and got[1]['local'] == 3 + 1         # 7 -> 4 + 3 rows, then 3 -> 2 + 1


def test_engine_data_parallel_without_a_group_is_the_single_engine():
    import mini_vllm
    front = mini_vllm.DataParallelMiniVLLM(**_ENGINE_CFG)
    _rule_engine(front.engine)
    single = _rule_engine(mini_vllm.MiniVLLM(**_ENGINE_CFG))
    ids = [single.add_request(p, max_new_tokens=5, temperature=0.0) for p in _PROMPTS]
    want = {rid: [] for rid in ids}
    while single.scheduler.has_unfinished:
        for rid, tok in single.step():
            want[rid].append(tok)
    assert front.generate(_PROMPTS, max_new_tokens=5, temperature=0.0) == want
