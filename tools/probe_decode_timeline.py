"""Timeline of one paged-decode launch, workgroup by workgroup: needs a side build of the library
with -DMI355_DECODE_TIMING=1 (thread 0 of every workgroup writes the 100 MHz clock and the shader
clock at entry, behind the block-table barrier, after its first pipeline step, after its last one
and after the merge of its sub-ranges into the workspace), selected through MI355_PAGED_LIB.
    MI355_PAGED_LIB=tools/_ab/lib_pd_timing.so python tools/probe_decode_timeline.py [metric|c2]
Launches are queued back to back behind a spin kernel (as in the benchmark's roofline groups); the
last few are decoded.  Times are microseconds on the 100 MHz clock (10 ns steps), relative to the
first workgroup's entry; cycle counts come from the shader clock of the workgroup itself."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vllm-from-scratch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402
from bench import CONFIGS, build_inputs  # noqa: E402
from mi355_paged import _lib, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
cfg = CONFIGS[name]
dev = torch.device('cuda', 0)
inp = build_inputs(cfg, dev, 0, sets=4 if name == 'c2' else 2)
B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
out = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
lib = _lib.load()
NWG = B * H                       # one kv head per workgroup in both benchmark configs
N = 12
stamps = [torch.zeros((NWG, 16), dtype=torch.int64, device=dev) for _ in range(N)]
stream = torch.cuda.current_stream().cuda_stream


def launch(i):
    kc, vc, bt = inp['pools'][i % len(inp['pools'])]
    rc = lib.mi355_paged_decode(out.data_ptr(), inp['q'].data_ptr(), H * D, kc.data_ptr(), vc.data_ptr(), bt.data_ptr(),
                                inp['ctx'].data_ptr(), B, H, H, D, BS, bt.shape[1], kc.shape[0], S, kc.stride(0),
                                kc.stride(1), kc.stride(2), 1.0 / D ** 0.5, ops.dtype_code(inp['q']),
                                stamps[i].data_ptr(), stamps[i].numel() * 8, stream)
    _lib.check(rc, 'mi355_paged_decode')


for i in range(4):
    launch(i)
torch.cuda.synchronize()
torch.cuda._sleep(20_000_000)
for i in range(N):
    launch(i)
torch.cuda.synchronize()
NAMES = ['entry', 'table staged', 'first step done', 'stream done', 'sub-ranges merged']
for i in range(N - 3, N):
    st = stamps[i].cpu().numpy()
    st = st[st[:, 0] != 0]                 # the launch's workgroups (the plan decides how many)
    NWG = len(st)
    if os.environ.get('PROBE_DUMP'):
        print(st[:3])
    if NWG == 0 or not st[:, 1:10:2].all():
        sys.exit(f"stamps missing ({int((st[:, :10] != 0).sum())} of {NWG * 10} written; library {_lib.LIB_PATH}): is "
                 "MI355_PAGED_LIB a -DMI355_DECODE_TIMING=1 build, and the plan one head per workgroup?")
    cyc = st[:, 1::2][:, :5].astype(np.float64)
    if st[:, 0:10:2].all():
        real = st[:, 0::2][:, :5].astype(np.float64) / 100.0      # us on the 100 MHz clock
    else:                                                         # that clock reads 0 here: the shader clock, at CLOCK_MHZ
        real = cyc / float(os.environ.get('CLOCK_MHZ', '100'))
    t0 = real[:, 0].min()
    prev = stamps[i - 1].cpu().numpy()
    prev = prev[prev[:, 0] != 0]
    prev_end = (prev[:, 8].astype(np.float64).max() / 100.0 if prev[:, 8].all()
                else prev[:, 9].astype(np.float64).max() / float(os.environ.get('CLOCK_MHZ', '100')))
    print(f"launch {i}: first entry {t0 - prev_end:+.2f} us after the previous launch's last workgroup finished its stream")
    for k, nm in enumerate(NAMES):
        r = real[:, k] - t0
        print(f"  {nm:>18}: first {r.min():6.2f}  median {np.median(r):6.2f}  p90 {np.percentile(r, 90):6.2f}  last {r.max():6.2f} us")
    life = real[:, 4] - real[:, 0]
    print(f"  workgroup lifetime: median {np.median(life):.2f} us, p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f};"
          f"  entry -> table {np.median(cyc[:, 1] - cyc[:, 0]):.0f} cycles, table -> first step {np.median(cyc[:, 2] - cyc[:, 1]):.0f},"
          f" stream {np.median(cyc[:, 3] - cyc[:, 2]):.0f}, merge {np.median(cyc[:, 4] - cyc[:, 3]):.0f}")
    started = np.sort(real[:, 0] - t0)
    print(f"  workgroups entered by 1 / 2 / 3 / 5 / 10 us: " + " / ".join(str(int((started <= t).sum())) for t in (1, 2, 3, 5, 10))
          + f" of {NWG};  still streaming at (last end - 1 / 2 / 3 us): "
          + " / ".join(str(int((real[:, 3] - t0 > (real[:, 3] - t0).max() - t).sum())) for t in (1, 2, 3)))
