"""The first MLA call of a process captured into a HIP graph (the 128-heads-per-workgroup kernel sets its
LDS size attribute on its first launch on a device: that must be legal under capture), replayed, and compared
with the 64-head kernel.  Run in a fresh process: python tools/probe_capture_first_call.py"""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'vllm-from-scratch_amd'))
import torch
from mi355_paged import ops
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
rows, H, S, R, P = 4, 128, 600, 512, 64
q = (torch.randn((rows, H, R + P), generator=g, device=dev) * 0.3).bfloat16()
c = torch.randn((rows, S, R), generator=g, device=dev).bfloat16()
k = torch.randn((rows, S, P), generator=g, device=dev).bfloat16()
pos = torch.arange(S, device=dev, dtype=torch.float32)
inv = 1.0 / (10000.0 ** (torch.arange(0, P, 2, device=dev).float() / P))
ang = torch.outer(pos, inv)
cos, sin = torch.cat([ang, ang], -1).cos(), torch.cat([ang, ang], -1).sin()
ctx = torch.full((rows,), S, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):          # the first call of the process is a captured one
    out = ops.mla_decode(q, c, k, cos, sin, ctx, S, 0.07)
graph.replay()
torch.cuda.synchronize()
with ops.debug_knob('MI355_MLA_WIDE', 0):
    want = ops.mla_decode(q, c, k, cos, sin, ctx, S, 0.07)
print('max diff', (out - want).abs().max().item())
