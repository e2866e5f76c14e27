"""Wall time per part of MiniVLLM.step over the timed pass of tools/bench_engine_mixed.py's workload
(perf_counter around the methods; the sampler's read-back includes waiting for the device)."""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import numpy as np
import torch
import mini_vllm as mv

nreq, nseq, budget = ([int(x) for x in sys.argv[1:]] + [512, 128, 2048])[:3]
rng = np.random.default_rng(0)
plens = rng.integers(64, 513, size=nreq)
news = rng.integers(16, 65, size=nreq)
torch.manual_seed(0)
eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                  num_kv_blocks=nseq * 40, block_size=16, max_num_seqs=nseq, max_num_batched_tokens=budget,
                  device='cuda', dtype=torch.bfloat16)
acc, calls = defaultdict(float), defaultdict(int)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            calls[label] += 1
    setattr(obj, name, wrapper)


for rep in range(2):
    for p, n in zip(plens, news):
        eng.add_request([int(t) for t in rng.integers(1, 32000, size=int(p))], max_new_tokens=int(n), temperature=0.0)
    if rep:
        timed(eng.scheduler, 'schedule')
        timed(eng.scheduler, 'update')
        timed(eng.scheduler, 'schedule_decode_only')
        timed(eng.scheduler, 'update_decode_only')
        timed(eng, '_run_mixed')
        timed(eng, '_mixed_step')
        timed(eng, '_launch_decode')
        timed(eng, '_step_decode_only')
        timed(eng.sampler, 'sample', 'sampler.sample (waits for the device)')
        eng._device_tables.sync = eng._device_tables.sync
        timed(eng._device_tables, 'sync')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    while eng.scheduler.has_unfinished:
        eng.step()
        steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'{steps} steps, {dt * 1e3:.1f} ms')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f'  {k:45s} {v * 1e3:8.1f} ms  {calls[k]:5d} calls  {v / calls[k] * 1e6:8.1f} us each')
