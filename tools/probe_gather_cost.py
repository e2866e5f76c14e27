"""Where the per-step cost of the asynchronous id exchange comes from, on one GPU: the bench's decode
step alone; + an event recorded on the compute stream; + a side stream that waits for it and runs a
tiny copy; + the same with the compute stream later waiting for the side stream.
  python tools/probe_gather_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vllm-from-scratch_amd'))
import torch
import bench
from mi355_paged import ops

dev = torch.device('cuda', 0)
cfg = bench.CONFIGS['metric']
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
inp = bench.build_inputs(cfg, dev, seed=0, sets=2)
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
ids = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(4)]
recv = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(4)]
side = torch.cuda.Stream(device=dev)
events = [torch.cuda.Event() for _ in range(4)]
done = [torch.cuda.Event() for _ in range(4)]


def step(i, mode):
    kc, vc, bt = inp['pools'][i % 2]
    ops.kv_write_slots(kc, vc, inp['k_new'], inp['v_new'], inp['slots'][i % 2], 'NHSD')
    ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, layout='NHSD', out=out)
    ops.sample_argmax(inp['logits'], out=ids[i % 4])
    if mode >= 1:
        events[i % 4].record()
    if mode >= 2:
        side.wait_event(events[i % 4])
        with torch.cuda.stream(side):
            recv[i % 4].copy_(ids[i % 4], non_blocking=True)
            done[i % 4].record()
    if mode >= 3:
        torch.cuda.current_stream().wait_event(done[(i - 3) % 4])


for mode, name in ((0, 'step alone'), (1, '+ event record on the compute stream'),
                   (2, '+ side stream waits for it and copies the ids'),
                   (3, '+ compute stream waits for the copy of three steps ago')):
    for i in range(20):
        step(i, mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for i in range(n):
        step(i, mode)
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per step', flush=True)
