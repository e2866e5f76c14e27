"""Continuous batching through MiniVLLM: more requests than sequence slots, ragged prompts, so
prompt chunks and decode tokens share steps and finished requests are replaced from the waiting
queue.   python tools/bench_engine_mixed.py [requests max_num_seqs budget]"""
import os
import sys
import time

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

PASSES = int(os.environ.get('ENGINE_PASSES', '3'))
for rep in range(PASSES):                  # one engine serves all passes (fresh random prompts each): the first warms
    for p, n in zip(plens, news):          # kernels, GEMM plans and captures most step shapes, the later ones are timed
        eng.add_request([int(t) for t in rng.integers(1, 32000, size=int(p))], max_new_tokens=int(n), temperature=0.0)
    steps = fast = 0
    out_tokens = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while eng.scheduler.has_unfinished:
        before = eng._launches
        out_tokens += len(eng.step())
        fast += eng._launches > before
        steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep == 0:
        continue
    print(f"pass {rep + 1}: {len(eng._graphs)} captured decode shapes, {len(eng._mixed)} captured prompt-step shapes; ", end='')
    print(f"{nreq} requests (prompts U[64,512], {int(plens.sum())} prompt tokens; {out_tokens} generated), {nseq} slots, "
      f"budget {budget}: {dt * 1e3:.1f} ms, {steps} steps ({fast} captured decode launches), "
          f"{(int(plens.sum()) + out_tokens) / dt:.0f} tokens/s total, {out_tokens / dt:.0f} generated tokens/s", flush=True)
