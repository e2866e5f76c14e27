"""Host time to ISSUE each call of the bench's decode step (the queue is held by a spin kernel, so
nothing waits for the device):  python tools/probe_step_host_cost.py   (under torch.distributed.run
with BENCH_FORCE_DIST=1 it also times the id exchange through torch.distributed and the library)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vllm-from-scratch_amd'))
import torch
import bench
from mi355_paged import dp, ops

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
use_dist = bool(os.environ.get('BENCH_FORCE_DIST'))
if use_dist:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=dev)
cfg = bench.CONFIGS['metric']
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
inp = bench.build_inputs(cfg, dev, seed=0, sets=2)
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
ids = torch.empty((B,), dtype=torch.int32, device=dev)
kc, vc, bt = inp['pools'][0]
calls = {
    'kv_write_slots': lambda i: ops.kv_write_slots(kc, vc, inp['k_new'], inp['v_new'], inp['slots'][0], 'NHSD'),
    'paged_decode': lambda i: ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, layout='NHSD', out=out),
    'sample_argmax': lambda i: ops.sample_argmax(inp['logits'], out=ids),
}
if use_dist:
    native = dp.NativeComm(dev)
    gathers = []
    for label, kw in (('torch.distributed', {}), ('mi355_dp_allgather_i32', {'native': native})):
        g = dp.OverlappedTokenGather(B, dev, single_rank_collective=True, **kw)
        gathers.append(g)
        calls[f'exchange via {label}'] = lambda i, g=g: (g.send_buffer(i), g.launch(i))
for name, fn in calls.items():
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)          # ~0.1 s: the device is busy while the host issues
    n = 100
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f'{name}: {host:.1f} us of host time per call', flush=True)
if use_dist:
    for g in gathers:
        g.drain()
    torch.cuda.synchronize()
    native.close()
    dist.destroy_process_group()
