"""End-to-end MiniVLLM throughput on the GPU (random weights):
python tools/bench_engine.py [nseq prompt_len new_tokens [token_budget]]   (a budget below the prompt length chunks the prompts)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import mini_vllm as mv

nseq, plen, ntok, budget = ([int(x) for x in sys.argv[1:]] + [128, 256, 32, 8192][len(sys.argv) - 1:])[:4]
torch.manual_seed(0)
eng = mv.MiniVLLM(vocab_size=32000, hidden_size=512, num_layers=4, num_heads=8, intermediate_size=1536,
                  num_kv_blocks=nseq * ((plen + ntok) // 16 + 2), block_size=16, max_num_seqs=nseq,
                  max_num_batched_tokens=budget, device='cuda', dtype=torch.bfloat16,
                  fused_decode_step=os.environ.get('ENGINE_FUSED_DECODE', '1') != '0')
g = torch.Generator().manual_seed(1)
for _ in range(2):   # first pass warms up kernels / allocator
    for i in range(nseq):
        eng.add_request(torch.randint(1, 32000, (plen,), generator=g).tolist(), max_new_tokens=ntok, temperature=0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = prefill_steps = 0
    t_decode = None
    produced = 0
    while eng.scheduler.has_unfinished:
        out = eng.step()
        steps += 1
        produced += len(out)
        if t_decode is None and all(not r.is_prefill for r in eng.scheduler.running) and not eng.scheduler.waiting:
            torch.cuda.synchronize()
            t_decode = time.perf_counter()
            prefill_steps = steps
            produced_at = produced
    torch.cuda.synchronize()
    t1 = time.perf_counter()
dec_steps = steps - prefill_steps
print(f"{nseq} seqs x ({plen} prompt + {ntok} new): total {t1 - t0:.3f} s, {produced / (t1 - t0):.0f} tok/s; "
      f"prefill {prefill_steps} steps {t_decode - t0:.3f} s ({nseq * plen / (t_decode - t0):.0f} prompt tok/s); "
      f"decode {dec_steps} steps, {(t1 - t_decode) / max(dec_steps, 1) * 1e3:.3f} ms/step, "
      f"{(produced - produced_at) / (t1 - t_decode):.0f} tok/s")
