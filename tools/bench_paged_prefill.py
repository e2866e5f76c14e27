"""Time mi355_paged_prefill (new query tokens of a chunk attending to the paged cache).
Usage: python tools/bench_paged_prefill.py [nseg chunk pos0 Hq Hkv D [dtype]]   (dtype: bfloat16 | float16 | float32)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

nseg, chunk, pos0, Hq, Hkv, D = ([int(x) for x in sys.argv[1:7]] + [8, 1024, 3072, 32, 8, 128])[:6]
dtype = getattr(torch, sys.argv[7]) if len(sys.argv) > 7 else torch.bfloat16
BS = 16
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
ctx = pos0 + chunk
per = -(-ctx // BS)
NB = nseg * per + 4
kc = torch.randn((NB, BS, Hkv, D), generator=g, device=dev).to(dtype)      # engine layout [blocks, BS, heads, D]
vc = torch.randn((NB, BS, Hkv, D), generator=g, device=dev).to(dtype)
bt = torch.randperm(NB, generator=g, device=dev)[:nseg * per].view(nseg, per).to(torch.int32).contiguous()
T = nseg * chunk
q = torch.randn((T, Hq, D), generator=g, device=dev).to(dtype)
start = torch.arange(0, T, chunk, dtype=torch.int32, device=dev)
lens = torch.full((nseg,), chunk, dtype=torch.int32, device=dev)
p0 = torch.full((nseg,), pos0, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
# every query token i of a chunk sees pos0 + i + 1 keys
flops = 4.0 * Hq * D * nseg * sum(pos0 + i + 1 for i in range(chunk))
warm_card()
for _ in range(3):
    ops.paged_prefill(q, kc, vc, bt, start, lens, p0, chunk, ctx, layout='NSHD', out=out)
n = 20
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    ops.paged_prefill(q, kc, vc, bt, start, lens, p0, chunk, ctx, layout='NSHD', out=out)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"paged_prefill nseg={nseg} chunk={chunk} pos0={pos0} Hq={Hq} Hkv={Hkv} D={D}: {ms:.3f} ms "
      f"{flops / ms / 1e9:.1f} TFLOP/s" + (f" ({flops / ms / 1e9 / 25:.1f}% of 2.5 PF dense bf16)" if dtype != torch.float32 else " (fp32)"))
