"""The reference's own engine benchmark, same configuration: benchmark_throughput() of
05_mini_vllm/mini_vllm.py:936-974 (vocab 1000, hidden 256, 4 layers, 8 heads, intermediate 1024,
256 KV blocks of 16, 16 sequence slots, 1024-token budget; 16 prompts of 10-50 random tokens,
generate(max_new_tokens=20) = temperature 1.0 sampling, fp32), and the blog's quoted shape
(BLOG_PART3_V2.md:1856-1866: heads=4, batch 4, prompt 50, 20 new tokens; "~400-500 tokens/s",
labelled illustrative, on an RTX PRO 6000).  Output tokens per second of generate(), measured as
the reference measures it (wall clock around generate), on a warm engine.
"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import mini_vllm as mv


def run(label, heads, prompts, dtype=None, reps=5):
    torch.manual_seed(0)
    eng = mv.MiniVLLM(vocab_size=1000, hidden_size=256, num_layers=4, num_heads=heads, intermediate_size=1024,
                      num_kv_blocks=256, block_size=16, max_num_seqs=16, max_num_batched_tokens=1024,
                      device='cuda', dtype=dtype)
    best = None
    for rep in range(reps + 1):             # the first pass builds kernels' plans and captures the steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outputs = eng.generate(prompts, max_new_tokens=20)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out_tokens = sum(len(v) for v in outputs.values())
        if rep and (best is None or out_tokens / dt > best[0]):
            best = (out_tokens / dt, out_tokens, dt)
    print(f"{label}: {len(prompts)} requests, {sum(len(p) for p in prompts)} prompt tokens, {best[1]} output tokens "
          f"in {best[2] * 1e3:.1f} ms -> {best[0]:.0f} output tok/s (best of {reps} warm passes)", flush=True)


random.seed(0)
shipped = [[random.randint(1, 999) for _ in range(random.randint(10, 50))] for _ in range(16)]
blog = [[random.randint(1, 999) for _ in range(50)] for _ in range(4)]
run("benchmark_throughput() config, fp32 (reference default)", 8, shipped)
run("benchmark_throughput() config, bf16", 8, shipped, torch.bfloat16)
run("blog config (heads=4, batch 4, prompt 50), fp32", 4, blog)
