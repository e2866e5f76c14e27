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
# Stamps at the end of the workspace, which a timing build makes 65 x 64 bytes larger: the epilogues of
# workgroups 0 .. 63 on the 100 MHz clock, then workgroup 0's step phases on the shader clock.
need = ops._lib.load().mi355_mla_decode_workspace_bytes(B, H, R, P, S, ops._DTYPE_CODE[torch.bfloat16])
ws = ops._workspaces[(dev, torch.cuda.current_stream(dev).cuda_stream)]
t = ws[need - 65 * 64:need].view(torch.int64).cpu().tolist()
ph = t[8 * 64:]
n = max(ph[3], 1)
print(f'steps {ph[3]}: per step  barrier {ph[0] / n:.0f}  score phase {ph[1] / n:.0f}  value phase {ph[2] / n:.0f}  total {(ph[0] + ph[1] + ph[2]) / n:.0f} shader-clock ticks')
rows = [t[8 * b:8 + 8 * b] for b in range(64)]
base = min(e[1] for e in rows)
shown = {}
for e in rows:
    split, row, other_in = e[0] & 255, (e[0] >> 8) & 0xffffffff, e[0] >> 40
    if row >= 6:
        continue
    head = f'row {row} split {split} (XCD {e[7]}): started {(e[6] - base) / 100:7.2f}  loop end {(e[1] - base) / 100:6.2f}'
    if e[3] == 0:      # wrote its rows (or its record for the merge kernel) and left
        print(f'{head}  rows stored +{(e[2] - e[1]) / 100:.2f}  acknowledged +{(e[5] - e[2]) / 100:.2f} us')
    else:
        what = 'found the other counted, no record' if other_in else 'record issued'
        print(f'{head}  {what} +{(e[2] - e[1]) / 100:.2f}  acknowledged +{(e[3] - e[2]) / 100:.2f}'
              f'  counted +{(e[4] - e[3]) / 100:.2f}  {"combined and stored" if e[5] - e[4] > 100 else "left"} +{(e[5] - e[4]) / 100:.2f} us')
