"""Per-step phase times of the 128-heads-per-workgroup MLA kernel (wave 0 of workgroup 0), from a
-DWIDE_TIMING=1 side build (tools/build_wide_variants.sh timing "-DWIDE_TIMING=1"):
MI355_PAGED_LIB=$PWD/tools/_ab/lib_wide_timing.so python tools/probe_wide_timeline.py [rows heads ctx]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

B, H, S = ([int(x) for x in sys.argv[1:4]] + [256, 128, 4096])[:3]
R, P = 512, 64
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
c = torch.randn((B, S, R), generator=g, device=dev).bfloat16()
k = torch.randn((B, S, P), generator=g, device=dev).bfloat16()
q = (torch.randn((B, H, R + P), generator=g, device=dev) * 0.3).bfloat16()
pos = torch.arange(S, device=dev, dtype=torch.float32)
inv = 1.0 / (10000.0 ** (torch.arange(0, P, 2, device=dev).float() / P))
ang = torch.outer(pos, inv)
cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
warm_card()
for _ in range(20):
    out = ops.mla_decode(q, c, k, cos, sin, ctx, S, 0.072)
torch.cuda.synchronize()
t = out.view(-1)[:8].view(torch.int64).cpu().tolist()
n = max(t[3], 1)
print(f'steps {t[3]}: per step  barrier {t[0] / n:.0f}  score phase {t[1] / n:.0f}  value phase {t[2] / n:.0f}  total {(t[0] + t[1] + t[2]) / n:.0f} shader-clock ticks')
t = out.view(-1)[:160].view(torch.int64).cpu().tolist()
base = min(x for x in t[9:80:8] if x > 0)
for b in range(8):     # the epilogues of workgroups 0 .. 7 on the 100 MHz clock, in microseconds from the earliest loop end
    e = t[8 + 8 * b:16 + 8 * b]
    print(f'workgroup {b}: row {e[0] >> 8} split {e[0] & 255}  loop end {(e[1] - base) / 100:.2f}  records issued +{(e[2] - e[1]) / 100:.2f}'
          f'  acknowledged +{(e[3] - e[2]) / 100:.2f}  counted +{(e[4] - e[3]) / 100:.2f}  done +{(e[5] - e[4]) / 100:.2f} us')
