"""Secondary measurements: ragged-context decode, sampler kernels.  python tools/bench_misc.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import CONFIGS, build_inputs
from mi355_paged import ops
from _clock import warm_card

dev = torch.device('cuda', 0)


def timeit(fn, n=50, warm=5):
    warm_card(0.2)
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


cfg = CONFIGS['metric']
inp = build_inputs(cfg, dev, 0)
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
ragged = torch.randint(S // 2, S + 1, (B,), generator=g, device=dev, dtype=torch.int32)
nbytes = int(ragged.sum().item()) * H * D * 2 * 2
i = [0]


def dec(ctx):
    kc, vc, bt = inp['pools'][i[0] % 2]
    i[0] += 1
    ops.paged_decode(inp['q'], kc, vc, bt, ctx, S, out=out)


for tgt in (256, 512, 1024, 2048):
    with ops.debug_knob('MI355_PAGED_TARGET_WGS', tgt):
        us = timeit(lambda: dec(ragged))
    print(f"ragged U[S/2,S] decode target_wgs={tgt}: {us:.1f} us -> {nbytes / us / 1e3:.0f} GB/s", flush=True)

for vocab in (1000, 32000, 128256):
    logits = torch.randn((B, vocab), generator=g, device=dev)
    temp = torch.full((B,), 0.8, device=dev)
    top_p = torch.full((B,), 0.9, device=dev)
    top_k = torch.full((B,), 50, dtype=torch.int32, device=dev)
    print(f"vocab {vocab}: argmax {timeit(lambda: ops.sample_argmax(logits)):.1f} us | "
          f"temperature only {timeit(lambda: ops.sample_topk_topp(logits, temp, None, None, seed=1)):.1f} us | "
          f"top-p {timeit(lambda: ops.sample_topk_topp(logits, temp, top_p, None, seed=1)):.1f} us | "
          f"top-k+top-p {timeit(lambda: ops.sample_topk_topp(logits, temp, top_p, top_k, seed=1)):.1f} us", flush=True)
