"""Decode-kernel bandwidth at the metric shape per KV dtype (bf16, fp16, fp32) and head_dim:
python tools/bench_dtypes.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

dev = torch.device('cuda', 0)
CASES = [  # dtype, B, S, H, D, BS
    (torch.bfloat16, 256, 2048, 8, 64, 16), (torch.float16, 256, 2048, 8, 64, 16), (torch.float32, 256, 2048, 8, 64, 16),
    (torch.float32, 128, 2048, 8, 128, 16), (torch.bfloat16, 128, 2048, 8, 128, 16), (torch.bfloat16, 256, 2048, 8, 32, 16),
    (torch.bfloat16, 64, 2048, 8, 256, 16), (torch.float32, 256, 2048, 4, 32, 4),
]
for dtype, B, S, H, D, BS in CASES:
    g = torch.Generator(device=dev).manual_seed(0)
    per = S // BS
    NB = B * per + 8
    pools = []
    for _ in range(2):
        kc = torch.randn((NB, H, BS, D), generator=g, device=dev).to(dtype)
        vc = torch.randn((NB, H, BS, D), generator=g, device=dev).to(dtype)
        bt = torch.randperm(NB, generator=g, device=dev)[:B * per].view(B, per).to(torch.int32).contiguous()
        pools.append((kc, vc, bt))
    q = torch.randn((B, H, D), generator=g, device=dev).to(dtype)
    ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    es = q.element_size()
    nbytes = B * S * H * D * 2 * es + 2 * B * H * D * es + 4 * B * per
    n = 30
    warm_card()
    for i in range(5):
        ops.paged_decode(q, *pools[i % 2], ctx, S, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        ops.paged_decode(q, *pools[i % 2], ctx, S, out=out)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"{str(dtype):15s} B={B} S={S} H={H} D={D} BS={BS}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s "
          f"({nbytes / us / 1e3 / 80:.1f}% of 8 TB/s)", flush=True)
    del pools
