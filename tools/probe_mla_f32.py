"""fp32 MLA kernel: is it waiting for memory?  The same launch with every row reading its own
sequence (HBM) and with all rows reading sequence 0 (9.4 MB: cache-resident after the first pass).
Usage: python tools/probe_mla_f32.py [rows heads ctx]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops

B, H, S = ([int(x) for x in sys.argv[1:4]] + [256, 16, 4096])[:3]
R, P = 512, 64
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
ckv = torch.randn((B, S, R), generator=g, device=dev)
kr = torch.randn((B, S, P), generator=g, device=dev)
q = torch.randn((B, H, R + P), generator=g, device=dev) * 0.3
pos = torch.arange(S, device=dev, dtype=torch.float32)
inv = 1.0 / (10000.0 ** (torch.arange(0, P, 2, device=dev).float() / P))
ang = torch.outer(pos, inv)
cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
for name, rs in (("own sequences (HBM)", None), ("all rows -> sequence 0 (cached)", torch.zeros(B, dtype=torch.int32, device=dev))):
    n = 20
    for _ in range(3):
        ops.mla_decode(q, ckv, kr, cos, sin, ctx, S, 0.072, row_seq=rs)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.mla_decode(q, ckv, kr, cos, sin, ctx, S, 0.072, row_seq=rs)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    flops = 2.0 * B * H * S * (R + P + R)
    print(f"mla fp32 rows={B} H={H} ctx={S} {name}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s", flush=True)
