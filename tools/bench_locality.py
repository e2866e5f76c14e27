"""Decode kernel at the metric shape with a scattered vs a sequential block table (how much of the
gap to the streaming ceiling is DRAM locality of 16 KiB blocks): python tools/bench_locality.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import CONFIGS, algorithmic_bytes, build_inputs
from mi355_paged import ops
from _clock import warm_card

cfg = CONFIGS['metric']
dev = torch.device('cuda', 0)
inp = build_inputs(cfg, dev, 0)
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
nb = algorithmic_bytes(B, S, H, H, D, BS, 2)
per = S // BS
tables = {'scattered': [p[2] for p in inp['pools']],
          'sequential': [torch.arange(B * per, dtype=torch.int32, device=dev).view(B, per)] * 2,
          'sequential, sequences interleaved': [torch.arange(B * per, dtype=torch.int32, device=dev).view(per, B).t().contiguous()] * 2}
for name, bts in tables.items():
    n = 60
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    warm_card()
    for i in range(n + 10):
        kc, vc, _ = inp['pools'][i % 2]
        if i >= 10:
            ea[i - 10].record()
        ops.paged_decode(inp['q'], kc, vc, bts[i % 2], inp['ctx'], S, out=out)
        if i >= 10:
            eb[i - 10].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(ea, eb))
    med = ms[len(ms) // 2]
    print(f"{name}: median {med * 1e3:.1f} us -> {nb / med / 1e6:.0f} GB/s ({nb / med / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
