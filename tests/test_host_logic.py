"""Host-side integer logic of the drop-in modules (CPU only): block pools, slot mapping and
the scheduler, replayed against traces captured from the reference, plus the reference's own
host-only test assertions restated (01_paged_attention/test_paged_attention.py:22-55,
05_mini_vllm/test_mini_vllm.py:19-176)."""
import pytest

from conftest import load_json
from mini_vllm import BlockAllocator, Request, Scheduler
from paged_attention import BLOCK_SIZE, KVCachePool, compute_slot_mapping


# ------------------------------------------------------------- golden traces (bit-exact)
def test_slot_mapping_golden():
    for c in load_json('g4_slot_mapping.json'):
        assert compute_slot_mapping(c['block_table'], c['seq_len'], c['block_size']) == c['slots']


def test_slot_mapping_short_table_raises():
    with pytest.raises(IndexError):
        compute_slot_mapping([3], seq_len=9, block_size=4)


def test_kv_cache_pool_trace():
    g = load_json('g5_allocator_traces.json')['pool']
    pool = KVCachePool(num_blocks=g['num_blocks'], block_size=4, num_kv_heads=1, head_dim=4, num_layers=1)
    for rec in g['trace']:
        if rec['op'] == 'alloc':
            if 'error' in rec:
                with pytest.raises(RuntimeError) as e:
                    pool.allocate(rec['arg'])
                assert str(e.value) == rec['error']
            else:
                assert pool.allocate(rec['arg']) == rec['result']
        else:
            pool.free(rec['arg'])
        assert list(pool.free_blocks) == rec['free_blocks']
        assert list(pool.ref_counts) == rec['ref_counts']


def test_block_allocator_trace():
    g = load_json('g5_allocator_traces.json')['allocator']
    alloc = BlockAllocator(num_blocks=g['num_blocks'], block_size=g['block_size'])
    for rec in g['trace']:
        if rec['op'] == 'alloc':
            got = alloc.allocate(rec['arg'][0]) if rec['arg'][0] is not None else alloc.allocate()
            assert got == rec['result']
        elif rec['op'] == 'free':
            alloc.free(rec['arg'][0])
        else:
            alloc.mark_cached(rec['arg'][0], rec['arg'][1])
        assert list(alloc.free_blocks.keys()) == rec['free_order']
        assert list(alloc.ref_count) == rec['ref_count']
        assert alloc.num_free == rec['num_free']
        assert {str(k): v for k, v in alloc.hash_to_block.items()} == rec['hash_to_block']


@pytest.mark.parametrize('name', sorted(load_json('g6_scheduler_traces.json')))
def test_scheduler_trace(name):
    g = load_json('g6_scheduler_traces.json')[name]
    cfg = g['config']
    alloc = BlockAllocator(num_blocks=cfg['num_blocks'], block_size=cfg['block_size'])
    sched = Scheduler(max_num_seqs=cfg['max_num_seqs'], max_num_batched_tokens=cfg['max_num_batched_tokens'],
                      block_size=cfg['block_size'], allocator=alloc, enable_chunked_prefill=cfg['chunked'])
    for r in g['requests']:
        sched.add_request(Request(r['id'], list(r['prompt']), r['max_new_tokens']))
    for step in g['steps']:
        batch = sched.schedule()
        got = [dict(id=s.request.request_id, tokens=list(s.tokens), positions=list(s.positions),
                    is_prefill=bool(s.is_prefill), slot_mapping=list(s.slot_mapping),
                    block_table=list(s.request.block_table)) for s in batch]
        assert got == step['scheduled']
        if batch:
            sched.update(batch, step['next_tokens'])
            assert [r.request_id for r in sched.finished] == step['finished']
        assert alloc.num_free == step['num_free']
        assert [r.request_id for r in sched.waiting] == step['waiting']
        assert [r.request_id for r in sched.running] == step['running']
    assert {r.request_id: list(r.output_token_ids) for r in sched.finished} == g['outputs']


# ------------------------------------------------------------- reference assertions, restated
@pytest.fixture
def small_pool():
    return KVCachePool(num_blocks=16, block_size=BLOCK_SIZE, num_kv_heads=4, head_dim=32, num_layers=2)


def test_pool_alloc_free_basic(small_pool):
    n_free = len(small_pool.free_blocks)
    blocks = small_pool.allocate(3)
    assert len(blocks) == 3 and len(small_pool.free_blocks) == n_free - 3
    small_pool.free(blocks)
    assert len(small_pool.free_blocks) == n_free
    assert small_pool.kv_cache.shape == (2, 2, 16, 4, BLOCK_SIZE, 32)


def test_pool_oom(small_pool):
    with pytest.raises(RuntimeError, match='OOM'):
        small_pool.allocate(999)


def test_slot_mapping_correctness():
    slots = compute_slot_mapping([3, 7], seq_len=6, block_size=4)
    assert (slots[0], slots[3], slots[4], slots[5]) == (12, 15, 28, 29)


class TestBlockAllocator:
    def test_counts(self):
        alloc = BlockAllocator(num_blocks=10, block_size=16)
        assert alloc.num_free == 10
        assert alloc.allocate() is not None
        assert alloc.num_free == 9

    def test_free_restores_count(self):
        alloc = BlockAllocator(num_blocks=5, block_size=8)
        blocks = [alloc.allocate() for _ in range(5)]
        assert alloc.num_free == 0
        alloc.free(blocks[0])
        assert alloc.num_free == 1

    def test_oom_returns_none(self):
        alloc = BlockAllocator(num_blocks=3, block_size=8)
        [alloc.allocate() for _ in range(3)]
        assert alloc.allocate() is None

    def test_prefix_cache_reuse(self):
        alloc = BlockAllocator(num_blocks=10, block_size=16)
        block_id = alloc.allocate()
        alloc.mark_cached(block_id, block_hash=42)
        alloc.free(block_id)
        assert alloc.allocate(block_hash=42) == block_id

    def test_all_blocks_unique_and_reallocatable(self):
        alloc = BlockAllocator(num_blocks=20, block_size=16)
        ids = [alloc.allocate() for _ in range(20)]
        assert None not in ids and len(set(ids)) == 20
        for b in ids:
            alloc.free(b)
        assert all(alloc.allocate() is not None for _ in range(20))


def make_scheduler(max_batched=64, chunked=True, num_blocks=64):
    alloc = BlockAllocator(num_blocks=num_blocks, block_size=16)
    return Scheduler(max_num_seqs=8, max_num_batched_tokens=max_batched, block_size=16, allocator=alloc,
                     enable_chunked_prefill=chunked), alloc


class TestScheduler:
    def test_empty_schedule(self):
        sched, _ = make_scheduler()
        assert sched.schedule() == []
        assert not sched.has_unfinished

    def test_single_short_prefill(self):
        sched, _ = make_scheduler(max_batched=64)
        sched.add_request(Request("r0", [1, 2, 3], max_new_tokens=5))
        scheduled = sched.schedule()
        assert len(scheduled) == 1 and scheduled[0].tokens == [1, 2, 3] and scheduled[0].is_prefill

    def test_token_budget_respected(self):
        sched, _ = make_scheduler(max_batched=32, chunked=True)
        sched.add_request(Request("r0", list(range(64)), max_new_tokens=5))
        assert sum(len(s.tokens) for s in sched.schedule()) <= 32

    def test_chunked_prefill_progression(self):
        sched, _ = make_scheduler(max_batched=16, chunked=True)
        sched.add_request(Request("r0", list(range(64)), max_new_tokens=0))
        steps = 0
        while sched.has_unfinished and steps < 20:
            scheduled = sched.schedule()
            if not scheduled:
                break
            for s in scheduled:
                s.request.num_computed_tokens += len(s.tokens)
                if not s.request.is_prefill:
                    s.request.is_finished = True
            sched.running = [r for r in sched.running if not r.is_finished]
            steps += 1
        assert steps >= 4

    def test_max_num_seqs_limit(self):
        alloc = BlockAllocator(num_blocks=128, block_size=16)
        sched = Scheduler(max_num_seqs=2, max_num_batched_tokens=256, block_size=16, allocator=alloc)
        for i in range(5):
            sched.add_request(Request(f"r{i}", [1, 2, 3], max_new_tokens=5))
        assert len(sched.schedule()) <= 2

    def test_slot_mapping_valid(self):
        alloc = BlockAllocator(num_blocks=16, block_size=8)
        sched = Scheduler(max_num_seqs=4, max_num_batched_tokens=64, block_size=8, allocator=alloc)
        sched.add_request(Request("r0", list(range(16)), max_new_tokens=5))
        limit = alloc.num_blocks * alloc.block_size
        for s in sched.schedule():
            assert all(0 <= slot < limit for slot in s.slot_mapping)

    def test_multiple_requests_batched(self):
        sched, _ = make_scheduler(max_batched=256)
        for i in range(4):
            sched.add_request(Request(f"r{i}", [i + 1, i + 2, i + 3], max_new_tokens=5))
        assert len(sched.schedule()) == 4


def test_request_properties():
    r = Request("r", [4, 5, 6], 7)
    assert (r.prompt_len, r.num_tokens, r.is_prefill, r.next_tokens_to_compute) == (3, 3, True, [4, 5, 6])
    assert (r.temperature, r.top_p) == (1.0, 1.0)
    r.num_computed_tokens = 3
    r.output_token_ids.append(9)
    assert (r.num_tokens, r.is_prefill, r.next_tokens_to_compute) == (4, False, [9])
