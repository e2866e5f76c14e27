"""cProfile of the timed pass of tools/bench_engine_mixed.py (host time per function)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import numpy as np
import torch
import mini_vllm as mv

nreq, nseq, budget = ([int(x) for x in sys.argv[1:]] + [256, 128, 2048])[:3]
rng = np.random.default_rng(0)
plens = rng.integers(64, 513, size=nreq)
news = rng.integers(16, 65, size=nreq)
torch.manual_seed(0)
eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                  num_kv_blocks=nseq * 40, block_size=16, max_num_seqs=nseq, max_num_batched_tokens=budget,
                  device='cuda', dtype=torch.bfloat16)
for rep in range(2):
    for p, n in zip(plens, news):
        eng.add_request([int(t) for t in rng.integers(1, 32000, size=int(p))], max_new_tokens=int(n), temperature=0.0)
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep:
        pr.enable()
    steps = 0
    while eng.scheduler.has_unfinished:
        eng.step()
        steps += 1
    torch.cuda.synchronize()
    pr.disable()
    print(f'pass {rep}: {steps} steps, {(time.perf_counter() - t0) * 1e3:.1f} ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
pstats.Stats(pr).sort_stats('cumtime').print_stats(28)
