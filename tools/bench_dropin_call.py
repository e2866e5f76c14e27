"""Per-call time of the drop-in paged_attention.paged_attention_decode for one sequence: device-resident
tensors, a block table that already lives on the device, and host-resident tensors staged per call."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
import torch
import paged_attention as pa
dev='cuda'
H,D,BS,S=8,64,16,2048
pool=pa.KVCachePool(num_blocks=S//BS+4, block_size=BS, num_kv_heads=H, head_dim=D, num_layers=1, dtype=torch.bfloat16, device=dev)
bt=pool.allocate(S//BS)
pool.kv_cache.normal_()
q=torch.randn(H,D,device=dev,dtype=torch.bfloat16)
kv=pool.kv_cache[:,0]
for _ in range(20): out=pa.paged_attention_decode(q,kv,bt,S,D,H,H,BS)
torch.cuda.synchronize(); t0=time.perf_counter()
n=500
for _ in range(n): out=pa.paged_attention_decode(q,kv,bt,S,D,H,H,BS)
torch.cuda.synchronize(); t1=time.perf_counter()
print(f"paged_attention_decode, one sequence ctx {S}, device-resident: {(t1-t0)/n*1e6:.1f} us per call")
btt=torch.tensor(bt,dtype=torch.int32,device=dev)
t0=time.perf_counter()
for _ in range(n): out=pa.paged_attention_decode(q,kv,btt,S,D,H,H,BS)
torch.cuda.synchronize(); t1=time.perf_counter()
print(f"  with the block table already a device tensor: {(t1-t0)/n*1e6:.1f} us per call")
qc=q.cpu(); kvc=kv.cpu()
t0=time.perf_counter()
for _ in range(20): out=pa.paged_attention_decode(qc,kvc,bt,S,D,H,H,BS)
t1=time.perf_counter()
print(f"  host-resident tensors (staged through the device): {(t1-t0)/20*1e6:.1f} us per call")
