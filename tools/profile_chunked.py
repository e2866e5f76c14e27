"""cProfile of MiniVLLM steps with chunked prompts (host time per function)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import mini_vllm as mv
torch.manual_seed(0)
nseq, plen, ntok, budget = 8, 4096, 8, 1024
eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                  num_kv_blocks=nseq * ((plen + ntok) // 16 + 2), block_size=16, max_num_seqs=nseq,
                  max_num_batched_tokens=budget, device='cuda', dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1)
def fill():
    for i in range(nseq):
        eng.add_request(torch.randint(1, 32000, (plen,), generator=g).tolist(), max_new_tokens=ntok, temperature=0.0)
fill()
while eng.scheduler.has_unfinished: eng.step()
fill()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter(); n=0
while eng.scheduler.has_unfinished and n < 30:
    eng.step(); n+=1
torch.cuda.synchronize()
pr.disable()
print((time.perf_counter()-t0)/n*1e3, 'ms/step')
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
