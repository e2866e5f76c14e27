"""Time mi355_paged_decode alone (HIP events, rotating KV pools) over launch-plan settings.
Usage: [MI355_PAGED_LIB=path] python tools/tune_decode.py [metric|c2] [target_wgs ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from bench import CONFIGS, algorithmic_bytes, build_inputs
from mi355_paged import ops

name = sys.argv[1] if len(sys.argv) > 1 else 'metric'
targets = [int(x) for x in sys.argv[2:]] or [1024]
cfg = CONFIGS[name]
dev = torch.device('cuda', 0)
inp = build_inputs(cfg, dev, 0, sets=3 if name == 'c2' else 2)
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
nb = algorithmic_bytes(B, S, H, H, D, BS, 2)
for tgt in targets:
    ops._lib.load().mi355_debug_set_knob(b'MI355_PAGED_TARGET_WGS', tgt)
    n = 60
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n + 10):
        kc, vc, bt = inp['pools'][i % len(inp['pools'])]
        if i >= 10:
            ea[i - 10].record()
        ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, out=out)
        if i >= 10:
            eb[i - 10].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(ea, eb))
    med = ms[len(ms) // 2]
    print(f"{os.path.basename(os.environ.get('MI355_PAGED_LIB', 'default'))} {name} target_wgs={tgt}: "
          f"median {med * 1e3:.1f} us  min {ms[0] * 1e3:.1f} us  -> {nb / med / 1e6:.0f} GB/s "
          f"({nb / med / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
