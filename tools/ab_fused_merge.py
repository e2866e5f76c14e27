"""Split-token decode with the merge inside the launch against the two-launch form
(MI355_DECODE_FUSED_MERGE=65535 / 0), same process, alternating:  python tools/ab_fused_merge.py [Hq Hkv D]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops

dev = torch.device('cuda', 0)
BS = 16
Hq, Hkv, D = ([int(x) for x in sys.argv[1:]] + [8, 8, 64])[:3]      # e.g. 32 8 128: the matrix-core kernel
for B, S in ((1, 2048), (1, 8192), (1, 32768), (1, 131072), (2, 8192), (4, 8192), (4, 32768), (8, 4096), (16, 8192), (16, 32768),
             (32, 4096)):
    g = torch.Generator(device=dev).manual_seed(0)
    per = S // BS
    NB = B * per + 8
    nbytes = B * S * Hkv * D * 2 * 2
    sets = max(2, min(8, int(600e6 // nbytes) + 1))
    pools = []
    for _ in range(sets):
        kc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
        vc = torch.randn((NB, Hkv, BS, D), generator=g, device=dev).to(torch.bfloat16)
        bt = torch.randperm(NB, generator=g, device=dev)[:B * per].view(B, per).to(torch.int32).contiguous()
        pools.append((kc, vc, bt))
    q = torch.randn((B, Hq, D), generator=g, device=dev).to(torch.bfloat16)
    ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    res = {0: [], 1: []}
    for rep in range(6):
        for fused in (0, 1):
            with ops.debug_knob('MI355_DECODE_FUSED_MERGE', 65535 * fused):
                for i in range(5):
                    ops.paged_decode(q, *pools[i % sets], ctx, S, out=out)
                n = 100
                torch.cuda._sleep(3_000_000)        # the host queues ahead of the device
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(n):
                    ops.paged_decode(q, *pools[i % sets], ctx, S, out=out)
                b.record()
                torch.cuda.synchronize()
                res[fused].append(a.elapsed_time(b) / n * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"Hq={Hq} Hkv={Hkv} D={D} B={B:3d} ctx={S:6d} ({nbytes / 1e6:7.1f} MB): two launches {med[0]:6.1f} us, one launch {med[1]:6.1f} us "
          f"({nbytes / med[1] / 1e3:5.0f} GB/s)", flush=True)
    del pools
