"""Decode over ragged contexts at the metric shape against the launch plan's workgroup target, next
to the uniform batch: what finer token splits (merged inside the launch) buy when lengths differ,
and what they cost when they do not.   python tools/bench_ragged.py [NHSD|NSHD]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import CONFIGS, build_inputs
from mi355_paged import ops
from _clock import warm_card

dev = torch.device('cuda', 0)
LAYOUT = (sys.argv[1:] + ['NHSD'])[0]
cfg = CONFIGS['metric']
inp = build_inputs(cfg, dev, 0)
if LAYOUT == 'NSHD':        # the same contents with the rows of all heads interleaved
    inp['pools'] = [(kc.permute(0, 2, 1, 3).contiguous(), vc.permute(0, 2, 1, 3).contiguous(), bt)
                    for kc, vc, bt in inp['pools']]
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
cases = {
    'uniform S': torch.full((B,), S, dtype=torch.int32, device=dev),
    'U[S/2, S]': torch.randint(S // 2, S + 1, (B,), generator=g, device=dev, dtype=torch.int32),
    'U[1, S]': torch.randint(1, S + 1, (B,), generator=g, device=dev, dtype=torch.int32),
    'exponential, mean S/4': torch.clamp((torch.empty((B,), device=dev).exponential_(4.0 / S, generator=g)).to(torch.int32) + 1, max=S),
}
step = [0]


def run(ctx, n=60):
    def once():
        kc, vc, bt = inp['pools'][step[0] % 2]
        step[0] += 1
        ops.paged_decode(inp['q'], kc, vc, bt, ctx, S, out=out, layout=LAYOUT)
    warm_card()
    for _ in range(5):
        once()
    torch.cuda._sleep(3_000_000)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, ctx in cases.items():
    nbytes = int(ctx.sum().item()) * H * D * 2 * 2
    row = []
    for tgt in (256, 512, 1024, 2048):
        with ops.debug_knob('MI355_PAGED_TARGET_WGS', tgt):
            us = run(ctx)
        row.append(f"{tgt}: {us:6.1f} us ({nbytes / us / 1e3:5.0f} GB/s)")
    print(f"{name:22s} {nbytes / 1e6:7.1f} MB | " + " | ".join(row), flush=True)
