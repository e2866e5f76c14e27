"""Time mi355_mla_decode at the MLA configuration of BASELINE.json (kv_lora_rank=512, rope 64, ctx 4096).
Usage: python tools/bench_mla.py [batch heads ctx [float32]]
MLA_PAGED=<block size>: the same rows in randomly placed blocks of a pool, through mi355_mla_decode_paged
(MLA_PAGED_SEQUENTIAL=1: the blocks in sequence order; MLA_PAGED_SPLIT=1: c_kv and k_rope rows in a pool
each instead of side by side in one row of R + P)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops

B, H, S = ([int(x) for x in sys.argv[1:4]] + [64, 16, 4096])[:3]
DT = getattr(torch, sys.argv[4]) if len(sys.argv) > 4 else torch.bfloat16
ES = 4 if DT == torch.float32 else 2
R, P = 512, 64
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
sets = 3
pools = [((torch.randn((B, S, R), generator=g, device=dev)).to(DT),
          (torch.randn((B, S, P), generator=g, device=dev)).to(DT)) for _ in range(sets)]
if os.environ.get('MLA_DENSE_INTERLEAVED'):     # the dense call on rows [c_kv | k_rope] of one tensor, as a pool holds them
    pools = [(lambda both: (both[..., :R], both[..., R:]))(torch.cat((c, k), -1)) for c, k in pools]
BS = int(os.environ.get('MLA_PAGED', '0'))
if BS:     # one pool [blocks, BS, R + P] per set, the sequences' blocks in a random permutation
    per = (S + BS - 1) // BS
    paged = []
    for c, k in pools:
        order = torch.randperm(B * per, generator=g, device=dev)
        if os.environ.get('MLA_PAGED_SEQUENTIAL'):      # blocks in sequence order: what the addressing alone costs
            order = torch.arange(B * per, device=dev)
        pool = torch.empty((B * per, BS, R + P), dtype=DT, device=dev)
        rows = torch.cat((c, k), -1)
        rows = torch.nn.functional.pad(rows, (0, 0, 0, per * BS - S)).view(B * per, BS, R + P)
        pool[order] = rows
        if os.environ.get('MLA_PAGED_SPLIT'):
            pool = (pool[..., :R].contiguous(), pool[..., R:].contiguous())
        else:
            pool = (pool[..., :R], pool[..., R:])
        paged.append((pool, order.view(B, per).to(torch.int32).contiguous()))
    del pools
q = (torch.randn((B, H, R + P), generator=g, device=dev) * 0.3).to(DT)
pos = torch.arange(S, device=dev, dtype=torch.float32)
inv = 1.0 / (10000.0 ** (torch.arange(0, P, 2, device=dev).float() / P))
ang = torch.outer(pos, inv)
cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
nbytes = B * S * (R + P) * ES + B * H * (R + P) * ES + B * H * R * 4
flops = 2.0 * B * H * S * (R + P + R)


def call(i):
    if BS:
        pool, bt = paged[i % sets]
        ops.mla_decode_paged(q, pool[0], pool[1], bt, cos, sin, ctx, S, 0.072)
    else:
        c, k = pools[i % sets]
        ops.mla_decode(q, c, k, cos, sin, ctx, S, 0.072)


# The card idles at its lowest clocks and takes tens of milliseconds of load to leave them: a timing
# loop that starts cold reads up to 20 % slow for its first ~100 launches (217 us, then 187, then 182
# for three loops in a row in one process).  Half a second of the same work comes first.
WARM_S = float(os.environ.get('MLA_WARM_SECONDS', '0.5'))
import time
t_end = time.perf_counter() + WARM_S
i = 0
while time.perf_counter() < t_end:
    call(i)
    i += 1
    if i % 20 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
for tgt in [int(x) for x in os.environ.get('MLA_TARGETS', '0,1024,2048,0').split(',')]:
    ops._lib.load().mi355_debug_set_knob(b'MI355_MLA_TARGET_WGS', tgt)      # 0 = the library's own plan
    n = 60
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n + 5):
        if i >= 5:
            ev[i - 5][0].record()
        call(i)
        if i >= 5:
            ev[i - 5][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
    print(f"mla_decode{f' paged (blocks of {BS})' if BS else ''} {str(DT).split('.')[-1]} B={B} H={H} ctx={S} R={R} P={P} target={tgt or 'default'}: {ms * 1e3:.1f} us  "
          f"{nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 80:.1f}% of 8 TB/s)  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
