"""Host-side cost of F.linear on the GPU as the row count M changes from call to call.

Prints, for a set of distinct M, the host time of the first call with that M and of a repeat,
both with the device idle (synchronised before each call) so that only launch-side work
(library heuristics, kernel selection, code-object loading) is timed."""
import sys, time
import torch
import torch.nn.functional as F

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda:0')
w = torch.randn(out, hidden, device=dev, dtype=torch.bfloat16)
gen = torch.Generator().manual_seed(0)
ms = sorted(set(int(m) for m in torch.randint(64, 2049, (96,), generator=gen).tolist()))
F.linear(torch.randn(128, hidden, device=dev, dtype=torch.bfloat16), w)
torch.cuda.synchronize()


def host_us(x):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F.linear(x, w)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e6


first, again, third = [], [], []
xs = {m: torch.randn(m, hidden, device=dev, dtype=torch.bfloat16) for m in ms}
for m in ms:
    first.append(host_us(xs[m]))
for m in ms:
    again.append(host_us(xs[m]))
for m in ms:
    third.append(host_us(xs[m]))
for name, v in (('first call', first), ('second pass', again), ('third pass', third)):
    v = sorted(v)
    print(f'{name}: {len(v)} distinct M, host us median {v[len(v)//2]:.1f}, mean {sum(v)/len(v):.1f}, max {v[-1]:.1f}')
fixed = [host_us(xs[ms[0]]) for _ in range(50)]
print(f'one M repeated: median {sorted(fixed)[25]:.1f} us')
print('preferred blas:', torch.backends.cuda.preferred_blas_library())
