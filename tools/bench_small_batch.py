"""Decode-kernel bandwidth at small batches and long contexts (token-split path + merge):
python tools/bench_small_batch.py
The device-side cost per call: a spin kernel holds the stream while the host queues the calls (a
call costs the host 15-20 us, more than these launches take), so the events span the launches only.
SMALL_BATCH_SPLITS="a,b,..": also with MI355_DECODE_FUSED_MERGE set to each value."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

dev = torch.device('cuda', 0)
BS = 16
for Hq, Hkv, D in ((8, 8, 64), (32, 8, 128)):
    for B, S in ((1, 8192), (1, 32768), (4, 8192), (4, 32768), (16, 8192), (16, 32768), (64, 1024), (32, 2048)):
        g = torch.Generator(device=dev).manual_seed(0)
        per = S // BS
        NB = B * per + 8
        nbytes = B * S * Hkv * D * 2 * 2 + 2 * B * Hq * D * 2 + 4 * B * per
        sets = max(2, min(8, int(600e6 // nbytes) + 1))          # rotate enough pools to defeat the 256 MiB cache
        pools = []
        for _ in range(sets):
            kc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
            vc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
            bt = torch.randperm(NB, generator=g, device=dev)[:B * per].view(B, per).to(torch.int32).contiguous()
            pools.append((kc, vc, bt))
        q = torch.randn((B, Hq, D), generator=g, device=dev).to(torch.bfloat16)
        ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
        out = torch.empty_like(q)
        n = 40
        warm_card(0.3)
        for i in range(6):
            ops.paged_decode(q, *pools[i % sets], ctx, S, out=out)
        for fuse in [None] + [int(x) for x in os.environ.get('SMALL_BATCH_SPLITS', '').split(',') if x]:
            if fuse is not None:
                ops._lib.load().mi355_debug_set_knob(b'MI355_DECODE_FUSED_MERGE', fuse)
                ops.paged_decode(q, *pools[0], ctx, S, out=out)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(4_000_000)     # ~2 ms: the n calls are queued behind it
            a.record()
            for i in range(n):
                ops.paged_decode(q, *pools[i % sets], ctx, S, out=out)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / n * 1e3
            print(f"Hq={Hq} Hkv={Hkv} D={D} B={B:3d} ctx={S:6d} ({nbytes / 1e6:7.1f} MB)"
                  + (f" fused merge up to {fuse} splits" if fuse is not None else "") + f": {us:7.1f} us  "
                  f"{nbytes / us / 1e3:6.0f} GB/s ({nbytes / us / 1e3 / 80:.1f}% of 8 TB/s)", flush=True)
        ops._lib.load().mi355_debug_set_knob(b'MI355_DECODE_FUSED_MERGE', -2 ** 31)
        del pools
