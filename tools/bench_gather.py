"""Cost of the per-step id all-gather on the RCCL backend, alone and beside the decode step.
Run under torch.distributed.run (one process per GPU; one rank is enough for the host-side cost):
  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/bench_gather.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import torch.distributed as dist

dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
from mi355_paged import dp

B, n = 256, 300
g = dp.OverlappedTokenGather(B, dev, single_rank_collective=True)
for i in range(20):
    g.send_buffer(i).fill_(i)
    g.launch(i)
g.drain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    g.send_buffer(i)
    g.launch(i)
t1 = time.perf_counter()
g.drain()
torch.cuda.synchronize()
t2 = time.perf_counter()
if dist.get_rank() == 0:
    print(f"all-gather of {B} ids x {dist.get_world_size()} ranks: host issue {1e6 * (t1 - t0) / n:.1f} us/call, "
          f"end to end {1e6 * (t2 - t0) / n:.1f} us/call")
dist.destroy_process_group()
