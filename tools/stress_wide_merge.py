"""Random shapes through the 128-heads-per-workgroup MLA kernel with two token splits: merged inside the
launch (even splits) must equal the merge kernel's result bit for bit, the default (first split
shortened) within tolerance; dense and paged, ragged contexts, workspace contents scrambled.
python tools/stress_wide_merge.py [iterations [seed]]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import numpy as np
import torch
from mi355_paged import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)
R, P = 512, 64
fused_runs = 0
for it in range(iters):
    rows = int(rng.integers(1, 140))
    H = int(rng.choice([17, 24, 32, 40, 64, 65, 96, 128, 130, 200]))
    S = int(rng.integers(600, 5000))
    bs = int(rng.choice([0, 0, 8, 16, 64]))
    dtype = torch.bfloat16 if it % 2 else torch.float16
    g = torch.Generator(device=dev).manual_seed(it)
    nseq = min(rows, 6)                      # rows share a few cached sequences
    c_kv = torch.randn((nseq, S, R), generator=g, device=dev).to(dtype)
    k_rope = torch.randn((nseq, S, P), generator=g, device=dev).to(dtype)
    row_seq = torch.randint(0, nseq, (rows,), generator=g, device=dev).to(torch.int32)
    q = (torch.randn((rows, H, R + P), generator=g, device=dev) * 0.4).to(dtype)
    ctx = torch.randint(0, S + 1, (rows,), generator=g, device=dev).to(torch.int32)
    ctx[int(rng.integers(0, rows))] = S
    pos = torch.arange(S, device=dev, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, P, 2, device=dev).float() / P))
    ang = torch.outer(pos, inv)
    cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
    if bs:
        per = -(-S // bs)
        order = torch.randperm(nseq * per, generator=g, device=dev)
        pool = torch.zeros((nseq * per, bs, R + P), dtype=dtype, device=dev)
        pool[order] = torch.nn.functional.pad(torch.cat((c_kv, k_rope), -1), (0, 0, 0, per * bs - S)).view(nseq * per, bs, R + P)
        bt = order.view(nseq, per).to(torch.int32).contiguous()
        call = lambda: ops.mla_decode_paged(q, pool[..., :R], pool[..., R:], bt, cos, sin, ctx, S, 0.08, row_seq=row_seq)
    else:
        call = lambda: ops.mla_decode(q, c_kv, k_rope, cos, sin, ctx, S, 0.08, row_seq=row_seq)
    n_ht = (H + 127) // 128
    target = 2 * rows * n_ht                 # two splits
    code = ops._DTYPE_CODE[dtype]
    with ops.debug_knob('MI355_MLA_WIDE_SPLIT_TOKENS', 1), ops.debug_knob('MI355_MLA_TARGET_WGS', target):
        lib = ops._lib.load()
        need = lib.mi355_mla_decode_workspace_bytes(rows, H, R, P, S, code)
        with ops.debug_knob('MI355_MLA_WIDE_FUSED_MERGE', 0):
            plain = lib.mi355_mla_decode_workspace_bytes(rows, H, R, P, S, code)
            want = call()
        fused_runs += need == plain + rows * n_ht * 8
        ws, _ = ops._workspace(dev, need, torch.cuda.current_stream(dev).cuda_stream)
        if it % 3 == 0:
            ws.fill_(255)
        elif it % 3 == 1:
            ws.view(torch.int64)[:need // 8].random_(-2 ** 62, 2 ** 62, generator=g)
        with ops.debug_knob('MI355_MLA_WIDE_SKEW_TOKENS', 0):
            even = call()
        skewed = call()
    if not torch.equal(even.view(torch.int32), want.view(torch.int32)):
        sys.exit(f"iteration {it}: rows {rows} heads {H} S {S} block {bs} {dtype}: merged in the launch differs from the merge kernel")
    if not torch.allclose(skewed, want, rtol=2e-2, atol=2e-3) or torch.isnan(skewed).any():
        sys.exit(f"iteration {it}: rows {rows} heads {H} S {S} block {bs} {dtype}: shortened first split off by {(skewed - want).abs().max().item()}")
    if it % 20 == 19:
        print(f"{it + 1} shapes, {fused_runs} merged in the launch", flush=True)
print(f"ok: {iters} shapes, {fused_runs} of them merged in the launch")
