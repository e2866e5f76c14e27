"""Reference-shaped CPU decode, used only as the timed CPU baseline in bench.py.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product).

This is the same operation sequence the reference executes for one sequence
(01_paged_attention/paged_attention.py:85-99): for every kv head, index the
cache once per token through the block table, stack the rows, then
einsum -> softmax -> einsum in torch on the host cores.  It is what a user of
the reference pays on CPU, restated so it can run on the GPU host where the
reference itself is not present.  tests/test_oracle_golden.py checks it against
the same golden vectors as the numpy oracle.
"""
from __future__ import annotations

import math

import torch


def decode_one_sequence(query: torch.Tensor, kv_cache: torch.Tensor, block_table,
                        seq_len: int, block_size: int) -> torch.Tensor:
    """query [Hq, D]; kv_cache [2, NB, Hkv, BS, D]; returns [Hq, D]."""
    n_q, d = query.shape
    n_kv = kv_cache.shape[2]
    group = n_q // n_kv
    inv_sqrt_d = 1.0 / math.sqrt(d)
    result = torch.zeros_like(query)
    where = [(block_table[t // block_size], t % block_size) for t in range(seq_len)]
    for h in range(n_kv):
        k_rows = torch.stack([kv_cache[0, blk, h, row] for blk, row in where])
        v_rows = torch.stack([kv_cache[1, blk, h, row] for blk, row in where])
        qs = query[h * group:(h + 1) * group]
        weights = torch.softmax(torch.einsum('gd,sd->gs', qs, k_rows) * inv_sqrt_d, dim=-1)
        result[h * group:(h + 1) * group] = torch.einsum('gs,sd->gd', weights, v_rows)
    return result
