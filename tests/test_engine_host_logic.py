"""Host-side planning of the engine (CPU only): which kernel serves which token of a flattened
step, and the persistent device block-table mirror (exercised here on torch's CPU device)."""
import torch

from mini_vllm import DeviceBlockTables, Request, build_attn_plan


def _req(rid, table):
    r = Request(rid, [1], 1)
    r.block_table = list(table)
    return r


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
    assert plan.causal_index.tolist() == [4, 5] and plan.cu_seqlens.tolist() == [0, 2]
    assert plan.paged_index.tolist() == [0, 1, 2, 3]
    assert plan.paged_context.tolist() == [8, 5, 6, 7] and plan.max_context == 8
    assert plan.paged_tables.tolist() == [[3, 8], [5, 0], [5, 0], [5, 0]]      # padded with 0


def test_plan_reference_semantics_without_block_tables():
    """No block tables (or reference_compat): prompt chunks attend within the chunk only and a
    decode row reads flat slot 0, like 05_mini_vllm/mini_vllm.py:437-481."""
    for kwargs in (dict(block_tables=None), dict(block_tables=[[3], [4]], reference_compat=True)):
        plan = build_attn_plan([5, 9, 10], [1, 2], [False, True], kwargs.pop('block_tables'), 'cpu', **kwargs)
        assert plan.causal_index.tolist() == [1, 2] and plan.cu_seqlens.tolist() == [0, 2]
        assert plan.paged_index.tolist() == [0]
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
