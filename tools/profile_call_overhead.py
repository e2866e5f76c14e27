"""cProfile of repeated ops.paged_decode calls on a small problem: where the host time of one call goes."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
dev = torch.device('cuda', 0)
B,H,D,BS,S = 4,8,64,16,256
NB = B*S//BS
kc = torch.randn((NB,H,BS,D), device=dev).to(torch.bfloat16); vc = kc.clone()
bt = torch.arange(NB, dtype=torch.int32, device=dev).view(B,-1)
q = torch.randn((B,H,D), device=dev).to(torch.bfloat16); ctx = torch.full((B,), S, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
logits = torch.randn((B, 32000), device=dev)
for _ in range(100): ops.paged_decode(q,kc,vc,bt,ctx,S,out=out)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(2000): ops.paged_decode(q,kc,vc,bt,ctx,S,out=out)
t1=time.perf_counter(); torch.cuda.synchronize()
print(f"paged_decode host issue: {(t1-t0)/2000*1e6:.1f} us/call")
t0=time.perf_counter()
for _ in range(2000): ops.sample_argmax(logits)
t1=time.perf_counter(); torch.cuda.synchronize()
print(f"sample_argmax host issue: {(t1-t0)/2000*1e6:.1f} us/call")
x = torch.randn((B, 512), device=dev).to(torch.bfloat16); w = torch.ones(512, device=dev).to(torch.bfloat16)
t0=time.perf_counter()
for _ in range(2000): ops.rmsnorm(x, w, 1e-6)
t1=time.perf_counter(); torch.cuda.synchronize()
print(f"rmsnorm host issue: {(t1-t0)/2000*1e6:.1f} us/call")
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): ops.paged_decode(q,kc,vc,bt,ctx,S,out=out)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
