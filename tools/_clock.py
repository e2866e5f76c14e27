"""Clock warm-up shared by the timing scripts in this directory.

An idle MI355X sits at its lowest clocks and needs tens of milliseconds of load to leave them: a
timing loop that starts on a cold card reads up to 20 % slow for its first ~15 ms (the MLA decode
at 256 rows x 4096 tokens: 217 us, then 187, then 182 for three 65-launch loops in a row in one
process).  warm_card() keeps the card busy for half a second; the scripts call it right before
their own few warm-up calls of the kernel they time."""
import os
import time

import torch

_buf = {}


def warm_card(seconds: float = 0.5) -> None:
    seconds = float(os.environ.get('MI355_WARM_SECONDS', seconds))     # (counter passes shorten it)
    dev = torch.device('cuda', torch.cuda.current_device())
    if dev not in _buf:
        _buf[dev] = (torch.randn((4096, 4096), device=dev, dtype=torch.bfloat16),
                     torch.empty((64 << 20,), device=dev, dtype=torch.float32))
    a, big = _buf[dev]
    end = time.perf_counter() + seconds
    while time.perf_counter() < end:
        for _ in range(10):      # matrix cores and HBM both: the power state follows the load
            torch.mm(a, a)
            big.add_(1.0)
        torch.cuda.synchronize()
