"""Device-side cost of mi355_sample_argmax per launch: a spin kernel holds the stream while n
launches are queued, then (one event pair around all of them) / n = kernel + dispatch gap.
Usage: python tools/bench_argmax.py [B vocab]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

B, V = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 32000)
dev = torch.device('cuda', 0)
sets = [torch.randn((B, V), device=dev) for _ in range(12)]     # 12 x 32.8 MB: past the 256 MiB Infinity Cache
out = torch.empty((B,), dtype=torch.int32, device=dev)
n = 120
for threads in (0,):
    for rot in (len(sets), 1):
        warm_card()
        for i in range(10):
            ops.sample_argmax(sets[i % rot], out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(10_000_000)
        a.record()
        for i in range(n):
            ops.sample_argmax(sets[i % rot], out=out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / n
        print(f"argmax {B}x{V} fp32 threads={threads or 'plan'} {'rotating 12 inputs' if rot > 1 else 'same input'}: "
              f"{us:.2f} us per launch -> {B * V * 4 / us / 1e3:.0f} GB/s", flush=True)
