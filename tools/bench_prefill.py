"""Time mi355_prefill_causal (matrix-core kernel vs the VALU causal-mode kernel).
Usage: python tools/bench_prefill.py [nseq seqlen Hq Hkv D]   (BENCH_MFMA_ONLY=1 skips the VALU kernel)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

args = [int(x) for x in sys.argv[1:]] or [16, 2048, 8, 8, 64]
nseq, L, Hq, Hkv, D = args
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
T = nseq * L
q = torch.randn((T, Hq, D), generator=g, device=dev).to(torch.bfloat16)
k = torch.randn((T, Hkv, D), generator=g, device=dev).to(torch.bfloat16)
v = torch.randn((T, Hkv, D), generator=g, device=dev).to(torch.bfloat16)
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
flops = 4.0 * nseq * Hq * D * (L * (L + 1) / 2)          # QK^T + PV over the causal triangle

for name, env in (('mfma', None), ('valu', '1'))[:1 if os.environ.get('BENCH_MFMA_ONLY') else 2]:
    ops._lib.load().mi355_debug_set_knob(b'MI355_PREFILL_VALU', int(env or 0))
    n = 20 if name == 'mfma' else 5
    warm_card()
    for _ in range(3):
        ops.prefill_causal(q, k, v, cu, L)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.prefill_causal(q, k, v, cu, L)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"{name}: nseq={nseq} len={L} Hq={Hq} Hkv={Hkv} D={D}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s "
          f"({flops / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF dense bf16)", flush=True)
