"""Time mi355_paged_decode alone with the host running ahead of the GPU: a spin kernel holds the
stream while the timed launches (each between two HIP events, KV pools rotating) are queued, so
an event pair sees the device-side cost of a launch and not the host's call overhead.
Usage: [MI355_PAGED_LIB=path] python tools/tune_c2.py [metric|c2] [NAME=VALUE[,NAME=VALUE] ...]
Each argument after the config is one variant: launch-plan knobs (mi355_debug_set_knob) set for its launches."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import CONFIGS, algorithmic_bytes, build_inputs
from mi355_paged import ops

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
variants = sys.argv[2:] or ['']
cfg = CONFIGS[name]
dev = torch.device('cuda', 0)
inp = build_inputs(cfg, dev, 0, sets=4 if name == 'c2' else 2)
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
nb = algorithmic_bytes(B, S, H, H, D, BS, 2)
n = 100
for var in variants:
    knobs = [kv.split('=') for kv in filter(None, var.split(','))]
    for k, v in knobs:
        ops._lib.check(ops._lib.load().mi355_debug_set_knob(k.encode(), int(v)), 'mi355_debug_set_knob')
    for i in range(10):
        kc, vc, bt = inp['pools'][i % len(inp['pools'])]
        ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, out=out)
    torch.cuda.synchronize()
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda._sleep(20_000_000)
    for i in range(n):
        kc, vc, bt = inp['pools'][i % len(inp['pools'])]
        ea[i].record()
        ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, out=out)
        eb[i].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(ea, eb))
    # back to back, no events inside: (kernel + launch gap) per launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for i in range(n):
        kc, vc, bt = inp['pools'][i % len(inp['pools'])]
        ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, out=out)
    e1.record()
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) / n
    avg = sum(ms) / n
    print(f"{name} [{var or 'default'}]: events avg {avg * 1e3:.1f} us  p50 {ms[n // 2] * 1e3:.1f}  min {ms[0] * 1e3:.1f}"
          f"  p90 {ms[n * 9 // 10] * 1e3:.1f} -> {nb / avg / 1e6 / 80:.1f}% of 8 TB/s;  back-to-back {b2b * 1e3:.1f} us "
          f"({nb / b2b / 1e6 / 80:.1f}%)", flush=True)
    for k, _ in knobs:
        ops._lib.load().mi355_debug_set_knob(k.encode(), -2 ** 31)
