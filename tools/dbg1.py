import sys, torch
sys.path.insert(0, 'vllm-from-scratch_amd')
from mi355_paged import ops
dev = torch.device('cuda')
for dtype in (torch.float32, torch.bfloat16):
    H, D, BS, NB = 2, 32, 4, 3
    kc = torch.zeros((NB, H, BS, D), dtype=dtype, device=dev)
    vc = torch.arange(NB * H * BS * D, dtype=torch.float32, device=dev).view(NB, H, BS, D).to(dtype) / 64
    q = torch.ones((1, H, D), dtype=dtype, device=dev)
    bt = torch.tensor([[1, 0]], dtype=torch.int32, device=dev)
    for S in (1, 2, 5):
        cl = torch.tensor([S], dtype=torch.int32, device=dev)
        out = ops.paged_decode(q, kc, vc, bt, cl, S)
        torch.cuda.synchronize()
        rows = torch.stack([vc[bt[0, i // BS].item(), :, i % BS] for i in range(S)]).float().mean(0)
        print(dtype, S, 'got', out[0, 0, :8].float().cpu().tolist(), 'want', rows[0, :8].cpu().tolist())
