"""Where a steady-state decode step of MiniVLLM spends host time: python tools/profile_engine.py [nseq plen ntok]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import mini_vllm as mv

nseq, plen, ntok = ([int(x) for x in sys.argv[1:]] + [256, 512, 64])[:3]
torch.manual_seed(0)
eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                  num_kv_blocks=nseq * ((plen + ntok) // 16 + 2), block_size=16, max_num_seqs=nseq,
                  max_num_batched_tokens=nseq * plen, device='cuda', dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1)
for i in range(nseq):
    eng.add_request(torch.randint(1, 32000, (plen,), generator=g).tolist(), max_new_tokens=ntok, temperature=0.0)
eng.step()                      # whole prefill in one step
for _ in range(8):
    eng.step()                  # capture + warm
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    eng.step()
pr.disable()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step under cProfile, {nseq} seqs")
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
t0 = time.perf_counter()
for _ in range(n):
    eng.step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step plain")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
gr = next(reversed(eng._graphs.values()))
ev[0].record()
for _ in range(10):
    gr.graph.replay()
ev[1].record()
torch.cuda.synchronize()
print(f"graph replay alone: {ev[0].elapsed_time(ev[1]) / 10 * 1e3:.1f} us GPU time per step")
