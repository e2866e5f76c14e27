"""Decode with kv heads spread over more workgroups (MI355_PAGED_BALANCE_WGS) against the plain plan,
alternating in one process, uniform contexts:  python tools/ab_balance.py [NHSD|NSHD]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops

dev = torch.device('cuda', 0)
BS = 16
LAYOUT = (sys.argv[1:] + ['NHSD'])[0]          # NHSD = [blocks, heads, rows, d]; NSHD = [blocks, rows, heads, d]
for dtype, B, S, H, D in ((torch.bfloat16, 256, 2048, 8, 64), (torch.bfloat16, 256, 2048, 8, 32),
                          (torch.bfloat16, 128, 2048, 8, 128), (torch.float32, 256, 2048, 8, 64),
                          (torch.bfloat16, 256, 4096, 8, 64), (torch.bfloat16, 512, 1024, 8, 64),
                          (torch.bfloat16, 1024, 512, 8, 64), (torch.bfloat16, 256, 2048, 4, 128)):
    g = torch.Generator(device=dev).manual_seed(0)
    per = S // BS
    NB = B * per + 8
    pools = []
    for _ in range(2):
        shape = (NB, H, BS, D) if LAYOUT == 'NHSD' else (NB, BS, H, D)
        kc = torch.randn(shape, generator=g, device=dev).to(dtype)
        vc = torch.randn(shape, generator=g, device=dev).to(dtype)
        bt = torch.randperm(NB, generator=g, device=dev)[:B * per].view(B, per).to(torch.int32).contiguous()
        pools.append((kc, vc, bt))
    q = torch.randn((B, H, D), generator=g, device=dev).to(dtype)
    ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    nbytes = B * S * H * D * 2 * q.element_size()
    res = {}
    for bal in (0, 512, 1024, 2048):
        res[bal] = []
    for rep in range(5):
        for bal in res:
            with ops.debug_knob('MI355_PAGED_BALANCE_WGS', bal):
                for i in range(4):
                    ops.paged_decode(q, *pools[i % 2], ctx, S, out=out, layout=LAYOUT)
                n = 40
                torch.cuda._sleep(3_000_000)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(n):
                    ops.paged_decode(q, *pools[i % 2], ctx, S, out=out, layout=LAYOUT)
                b.record()
                torch.cuda.synchronize()
                res[bal].append(a.elapsed_time(b) / n * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{str(dtype)[6:]:9s} B={B:4d} S={S} H={H} D={D:3d}: " +
          " | ".join(f"{k}: {v:6.1f} us" for k, v in med.items()) + f"   ({nbytes / med[0] / 1e3:.0f} GB/s plain)", flush=True)
    del pools
