"""Where the host time of MiniVLLM prompt steps goes: python tools/profile_prefill_steps.py [nseq plen budget]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import mini_vllm as mv

nseq, plen, budget = ([int(x) for x in sys.argv[1:]] + [256, 512, 8192])[:3]


def build():
    torch.manual_seed(0)
    eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                      num_kv_blocks=nseq * (plen // 16 + 3), block_size=16, max_num_seqs=nseq,
                      max_num_batched_tokens=budget, device='cuda', dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    for i in range(nseq):
        eng.add_request(torch.randint(1, 32000, (plen,), generator=g).tolist(), max_new_tokens=4, temperature=0.0)
    return eng


eng = build()
while any(r.is_prefill for r in eng.scheduler.running + eng.scheduler.waiting):
    eng.step()                                  # warm-up pass over the same shapes
torch.cuda.synchronize()
eng = build()
steps = 0
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
while any(r.is_prefill for r in eng.scheduler.running + eng.scheduler.waiting):
    eng.step()
    steps += 1
pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} prompt steps, {dt / steps * 1e3:.2f} ms/step under cProfile, {nseq * plen / dt:.0f} prompt tok/s")
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
