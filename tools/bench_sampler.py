"""Sampler kernel timings by mode (HIP events around each call; run under rocprofv3 for kernel times).
Usage: python tools/bench_sampler.py [batch vocab [logit_std]]   (logit_std 1: flat rows; 4-6: peaked rows, as a model's)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
from _clock import warm_card

B, V = ([int(x) for x in sys.argv[1:3]] + [256, 32000])[:2]
STD = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(1)
logits = [torch.randn((B, V), generator=g, device=dev) * STD for _ in range(4)]
temp = torch.full((B,), 0.8, device=dev)
modes = {
    'argmax': lambda x: ops.sample_argmax(x),
    'temperature': lambda x: ops.sample_topk_topp(x, temp, None, None, seed=1),
    'top_p=0.9': lambda x: ops.sample_topk_topp(x, temp, torch.full((B,), 0.9, device=dev), None, seed=1),
    'top_p=0.5': lambda x: ops.sample_topk_topp(x, temp, torch.full((B,), 0.5, device=dev), None, seed=1),
    'top_k=50': lambda x: ops.sample_topk_topp(x, temp, None, torch.full((B,), 50, dtype=torch.int32, device=dev), seed=1),
    'top_k=50,top_p=0.9': lambda x: ops.sample_topk_topp(x, temp, torch.full((B,), 0.9, device=dev),
                                                          torch.full((B,), 50, dtype=torch.int32, device=dev), seed=1),
}
only = os.environ.get('SAMPLER_MODES')
for name, fn in modes.items():
    if only and name not in only.split(';'):
        continue
    warm_card(0.2)
    for i in range(5):
        fn(logits[i % 4])
    n = 40
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(n):
        fn(logits[i % 4])
    b.record()
    torch.cuda.synchronize()
    print(f"B={B} vocab={V}{'' if STD == 1.0 else f' std={STD:g}'} {name}: {a.elapsed_time(b) / n * 1e3:.1f} us per call", flush=True)
