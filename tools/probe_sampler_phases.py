"""Where sample_kernel's time goes: needs a side build of the library with -DSAMPLER_TIMING=1
(thread 0 of every workgroup writes the shader clock at the kernel's phase boundaries into the
out_probs buffer), selected through MI355_PAGED_LIB.  Prints, per sampling mode, the median over
rows of the cycles between consecutive stamps.
    MI355_PAGED_LIB=tools/_ab/lib_samp_t0.so python tools/probe_sampler_phases.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
from mi355_paged import ops  # noqa: E402

B, V = 256, 32000
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
logits = torch.randn((B, V), generator=g, device=dev)
temp = torch.ones(B, device=dev)
NAMES = {0: 'start', 1: 'loaded + max', 2: 'exp + sum', 3: 'L0 bins zeroed', 4: 'L0 tallied', 5: 'L0 scanned', 6: 'L0 cut known',
         7: 'L0 list built', 8: 'L1 zeroed', 9: 'L1 tallied', 10: 'L1 scanned', 11: 'L1 cut', 12: 'L1 end', 13: 'L2 zeroed',
         14: 'L2 tallied', 15: 'L2 scanned', 16: 'L2 cut', 17: 'L2 end', 18: 'search done', 19: 'ties done', 20: 'draw start',
         21: 'draw scanned', 22: 'picker known', 23: 'walk',
         24: 'list: counted', 25: 'list: normalised', 26: 'list: ordered', 27: 'list: picked'}
for label, tp, tk in (('top_k=50', 1.0, 50), ('top_p=0.9', 0.9, 0), ('top_p=0.5', 0.5, 0)):
    top_p = torch.full((B,), tp, device=dev)
    top_k = torch.full((B,), tk, dtype=torch.int32, device=dev)
    for _ in range(3):
        _, probs = ops.sample_topk_topp(logits, temp, top_p=top_p, top_k=top_k, seed=1, offset=0, return_probs=True)
    torch.cuda.synchronize()
    st = probs.view(torch.int64).view(-1)[:B * 32].view(B, 32).cpu().numpy()
    print(label)
    prev = 0
    order = [0, 1, 2, 24, 25, 26, 27] if np.median(st[:, 27] - st[:, 0]) > 0 and tk else sorted(n for n in NAMES if n < 24)
    for n in order:
        at = np.median(st[:, n] - st[:, 0])
        if n > 0 and not (0 < at < 5e5):       # a stamp this mode never wrote (the buffer is not cleared)
            continue
        d = np.median(st[:, n] - st[:, prev]) if n else 0
        print(f"  {NAMES[n]:>16}: +{d:8.0f} cycles   (at {np.median(st[:, n] - st[:, 0]):8.0f})")
        prev = n
