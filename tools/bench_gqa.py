"""Decode-kernel bandwidth for grouped-query shapes: python tools/bench_gqa.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

dev = torch.device('cuda', 0)
SHAPES = [  # B, S, Hq, Hkv, D, BS
    (64, 4096, 32, 8, 128, 16), (64, 4096, 64, 8, 128, 16), (128, 2048, 16, 2, 64, 16), (256, 2048, 8, 1, 64, 16),
    (64, 4096, 32, 32, 128, 16), (32, 8192, 8, 8, 128, 32), (256, 2048, 8, 8, 64, 16),
]
for B, S, Hq, Hkv, D, BS in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    per = S // BS
    NB = B * per + 8
    pools = []
    for _ in range(2):
        kc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
        vc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
        bt = torch.randperm(NB, generator=g, device=dev)[:B * per].view(B, per).to(torch.int32).contiguous()
        pools.append((kc, vc, bt))
    q = torch.randn((B, Hq, D), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    nbytes = B * S * Hkv * D * 2 * 2 + 2 * B * Hq * D * 2 + 4 * B * per
    n = 30
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    warm_card()
    for i in range(n + 5):
        kc, vc, bt = pools[i % 2]
        if i >= 5:
            ev[i - 5][0].record()
        ops.paged_decode(q, kc, vc, bt, ctx, S, out=out)
        if i >= 5:
            ev[i - 5][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
    print(f"B={B} S={S} Hq={Hq} Hkv={Hkv} D={D} BS={BS}: {ms * 1e3:.1f} us  {nbytes / ms / 1e6:.0f} GB/s "
          f"({nbytes / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    del pools
    torch.cuda.empty_cache()
