"""Causal prompt attention on the vector ALUs (fp32, or 16-bit at head dims the matrix-core kernel does not
cover): one query row per lane (fp32, head_dim 32 / 64), the decode-shaped kernel with row groups, and the
same with one query row per workgroup.
Usage: python tools/bench_prefill_fp32.py [nseq seqlen Hq Hkv D dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

args = sys.argv[1:]
nseq, L, Hq, Hkv, D = [int(x) for x in args[:5]] if len(args) >= 5 else [16, 2048, 8, 8, 64]
dtype = getattr(torch, args[5]) if len(args) > 5 else torch.float32
dev = torch.device('cuda', 0)
T = nseq * L
q = torch.randn(T, Hq, D, device=dev, dtype=dtype)
k = torch.randn(T, Hkv, D, device=dev, dtype=dtype)
v = torch.randn(T, Hkv, D, device=dev, dtype=dtype)
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
flops = 4.0 * nseq * Hq * D * L * (L + 1) / 2
outs = {}
for name, mfma, tiled, env in (('matrix cores (fp32 MFMA)', 1, '1', '1'), ('one query row per lane', 0, '1', '1'),
                               ('row groups of 8', 0, '0', '1'), ('one row per workgroup', 0, '0', '0')):
    if mfma and dtype != torch.float32:
        continue
    ops._lib.load().mi355_debug_set_knob(b'MI355_PREFILL_VALU', 0 if mfma else 1)
    ops._lib.load().mi355_debug_set_knob(b'MI355_PREFILL_F32_MFMA', mfma)
    ops._lib.load().mi355_debug_set_knob(b'MI355_PREFILL_F32_TILED', int(tiled))
    ops._lib.load().mi355_debug_set_knob(b'MI355_PREFILL_ROW_GROUPS', int(env))
    warm_card()
    for _ in range(2):
        outs[name] = ops.prefill_causal(q, k, v, cu, L)
    n = 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.prefill_causal(q, k, v, cu, L)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f'{name}: nseq={nseq} len={L} Hq={Hq} Hkv={Hkv} D={D} {dtype}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s', flush=True)
vals = list(outs.values())
print('max |difference| from the last form:', max((v.float() - vals[-1].float()).abs().max().item() for v in vals[:-1]))
