"""What replaying the bench step as a HIP graph costs, piece by piece (one-rank RCCL communicator):
eager step; the step's launches alone in a graph; with the previous step's all-gather in a parallel
branch; with it in line after the sampler; and the eager step with the collective replayed as a
graph of its own on a side stream.  us per step over 200 steps, host queued ahead where it can."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vllm-from-scratch_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mi355_paged import dp, ops  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'metric']
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
B, S = cfg['B'], cfg['S']
inp = bench.build_inputs(cfg, dev, 0, sets=4)
out = torch.empty((B, cfg['H'], cfg['D']), dtype=torch.bfloat16, device=dev)
comm = dp.NativeComm(dev)
gather = dp.OverlappedTokenGather(B, dev, single_rank_collective=True, native=comm)
side = gather.side
STEPS = 200


def body(k):
    kc, vc, bt = inp['pools'][k % 4]
    ops.kv_write_slots(kc, vc, inp['k_new'], inp['v_new'], inp['slots'][k % 4], 'NHSD')
    ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, layout='NHSD', out=out)
    ops.sample_argmax(inp['logits'], out=gather.send[k % 4])


def timed(fn, label):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        fn(i)
    torch.cuda.synchronize()
    print(f"{label:60s} {1e6 * (time.perf_counter() - t0) / STEPS:8.2f} us/step", flush=True)


def capture(fn):
    graphs = []
    main = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(main):
        for k in range(4):
            fn(k, torch.cuda.current_stream())
        main.synchronize()
        for k in range(4):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main, capture_error_mode='thread_local'):
                fn(k, torch.cuda.current_stream())
            graphs.append(g)
    torch.cuda.synchronize()
    return graphs


timed(body, "eager, no collective")

gs = capture(lambda k, cur: body(k))
timed(lambda i: gs[i % 4].replay(), "graph: the three launches")


def forked(k, cur):
    side.wait_stream(cur)
    comm.all_gather_i32(gather.send[(k - 1) % 4], gather.recv[(k - 1) % 4], stream=side)
    body(k)
    cur.wait_stream(side)


gs2 = capture(forked)
timed(lambda i: gs2[i % 4].replay(), "graph: launches + previous ids' all-gather in a branch")


def inline(k, cur):
    body(k)
    comm.all_gather_i32(gather.send[k % 4], gather.recv[k % 4], stream=cur)


gs3 = capture(inline)
timed(lambda i: gs3[i % 4].replay(), "graph: launches + all-gather in line")


def inline_first(k, cur):
    comm.all_gather_i32(gather.send[(k - 1) % 4], gather.recv[(k - 1) % 4], stream=cur)
    body(k)


gs4 = capture(inline_first)
timed(lambda i: gs4[i % 4].replay(), "graph: previous ids' all-gather in line, then the launches")

timed(lambda i: (body(i), comm.all_gather_i32(gather.send[i % 4], gather.recv[i % 4])),
      "eager: launches + all-gather on the compute stream")


def eager_prev(i):
    comm.all_gather_i32(gather.send[(i - 1) % 4], gather.recv[(i - 1) % 4])
    body(i)


timed(eager_prev, "eager: previous ids' all-gather on the compute stream, launches")
comm.close()
