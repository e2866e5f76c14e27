#!/usr/bin/env python3
"""Headline benchmark: paged-attention decode + greedy sampling on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json `metric`): block_size=16, batch=256 sequences per GPU,
context 2048 tokens each, 8 query heads = 8 kv heads, head_dim 64, bf16 KV cache
laid out [blocks, heads, 16, 64] with a fully scattered (random permutation)
block table; synthetic random data, all inputs resident in HBM before timing.

One step = one decode step of one attention layer for the whole batch:
  (1) write the new token's K/V rows into their slots   (mi355_kv_write_slots)
  (2) paged decode attention over the 2048-token context (mi355_paged_decode)
  (3) greedy sampling from fp32 logits [256, 32000]      (mi355_sample_argmax)
  (4) with N > 1: all-gather of the sampled ids across ranks (RCCL), one collective per step,
      asynchronous: ranks do not need each other's ids to go on decoding, so a collective
      overlaps the steps after it.  By default the library's own communicator runs it on a
      low-priority side stream behind a one-wave watcher kernel that returns once the step's ids
      are written (mi355_dp_exchange_*): the stream of steps (1)-(3) carries nothing for the
      exchange.  BENCH_EXCHANGE=native (event record + stream wait per step) and =torch
      (torch.distributed) select the earlier forms; `config.exchange` says which one ran.
      --gather-every k (not the default) sends the ids of k consecutive steps in one collective
      and says so in `config.workload`.  Every id is exchanged, and every collective has
      completed, before the timed region closes.
Before the --warmup steps the step runs untimed for about a quarter of a second
(`clock_warmup_steps`): an idle card sits at its lowest clocks for its first tens of milliseconds.
`value` = sequences (= tokens) decoded per second over all ranks; the timed region holds exactly
--steps steps and no instrumentation.  `roofline` prices step (2), the dominant kernel, against
HBM bandwidth: algorithmic bytes / the average launch duration by HIP events on the launch stream,
taken right after the timed region (so its size does not depend on --steps and its events cost
the tokens/s figure nothing): groups of 16 decode launches in a row between one event pair
(`avg_launch_ms`), and, reported beside it, ROOFLINE_LAUNCHES launches inside full steps with an
event pair each (`single_launch_brackets`, which also spans the event packets, about 2.3 us).
`roofline.traffic` is the counter-measured HBM bytes per launch from the newest
profiles/*_pmc_traffic.json, reported only when that file was collected with the very library
this process loaded (sha256 match), else null with a `traffic_stale` note.  `cpu_baseline` times
the reference-shaped CPU decode (oracle/loop_form.py) on a bounded sample of the same sequences.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'vllm-from-scratch_amd'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides: 8 TB/s; ~6.3 TB/s measured copy)

CONFIGS = {
    # BASELINE.json metric config
    'metric': dict(B=256, S=2048, H=8, D=64, BS=16, vocab=32000),
    # BASELINE.json configs[1]
    'c2': dict(B=64, S=1024, H=8, D=64, BS=16, vocab=32000),
}


def algorithmic_bytes(B, S, Hq, Hkv, D, BS, esize, append=False):
    """Bytes one decode launch must move: K and V rows, q and out, block table, context lens;
    with the cache write in the launch, also the new K/V rows read and stored."""
    nblk = (S + BS - 1) // BS
    new_rows = 2 * (2 * B * Hkv * D * esize) if append else 0
    return B * S * Hkv * D * 2 * esize + 2 * B * Hq * D * esize + 4 * B * nblk + 4 * B + new_rows


def build_inputs(cfg, device, seed, sets=2):
    """`sets` independent KV pools that the steps rotate through, so no step finds its
    working set in the 256 MiB Infinity Cache left by the previous one."""
    B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
    g = torch.Generator(device=device).manual_seed(seed)
    per = S // BS
    NB = B * per + 64
    pools = []
    for _ in range(sets):
        kc = torch.randn((NB, H, BS, D), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        vc = torch.randn((NB, H, BS, D), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        bt = torch.randperm(NB, generator=g, device=device)[:B * per].view(B, per).to(torch.int32).contiguous()
        pools.append((kc, vc, bt))
    q = torch.randn((B, H, D), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    k_new = torch.randn((B, H, D), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    v_new = torch.randn((B, H, D), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
    ctx = torch.full((B,), S, dtype=torch.int32, device=device)
    logits = torch.randn((B, cfg['vocab']), generator=g, device=device, dtype=torch.float32)
    # slot of the newest token (position S-1) of every sequence, per pool
    slots = [(bt[:, (S - 1) // BS].to(torch.int64) * BS + (S - 1) % BS).contiguous() for _, _, bt in pools]
    return dict(pools=pools, q=q, k_new=k_new, v_new=v_new, ctx=ctx, logits=logits, slots=slots)


def cpu_baseline(cfg, inp, budget_s=4.0, max_seqs=256):
    """Reference-shaped CPU decode on a sample of the benchmark's own sequences (fp32, as the
    reference runs it).  Returns the JSON object for the bench line."""
    from oracle import loop_form, paged_oracle
    B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
    kc, vc, bt = inp['pools'][0]
    per = S // BS
    n = min(max_seqs, B)
    blocks = bt[:n].reshape(-1).long()
    kv = torch.stack([kc[blocks], vc[blocks]]).float().cpu()          # compact pool [2, n*per, H, BS, D]
    tables = torch.arange(n * per).view(n, per).tolist()
    q = inp['q'][:n].float().cpu()
    threads = torch.get_num_threads()
    done, t0 = 0, time.perf_counter()
    while done < n:
        loop_form.decode_one_sequence(q[done], kv, tables[done], S, BS)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # vectorised numpy restatement, for scale
    kvn = kv.numpy()
    v0 = time.perf_counter()
    nv = min(4, n)
    for i in range(nv):
        paged_oracle.paged_decode(q[i].numpy(), kvn[0], kvn[1], tables[i], S, BS)
    vdt = (time.perf_counter() - v0) / nv
    return {"value": round(done / dt, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{done} of {B} sequences (ctx {S}, {H} heads, d {D}, fp32) through "
                      f"oracle/loop_form.py, {dt:.1f} s",
            "ms_per_sequence": round(1e3 * dt / done, 2),
            "vectorised_numpy_ms_per_sequence": round(1e3 * vdt, 2)}


# Launches of the dominant kernel timed between HIP events after the tokens/s region.
ROOFLINE_LAUNCHES = max(8, int(os.environ.get('BENCH_ROOFLINE_LAUNCHES', '64')))
# ... and groups of launches in a row, one event pair per group (see the roofline loop)
ROOFLINE_GROUP = 16
ROOFLINE_GROUPS = max(4, int(os.environ.get('BENCH_ROOFLINE_GROUPS', '8')))


def measured_stream_bandwidth(device, reps=10):
    """What this box's HBM delivers to plain streaming kernels, by HIP events: a device-to-device
    copy of 1 GiB (bytes read + bytes written per second) and a read-only pass (a sum) over it."""
    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    out = {}
    for name, fn, nbytes in (("copy", lambda: dst.copy_(src), 2 * n * 4), ("read", lambda: src.sum(), n * 4)):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
    del src, dst
    return out


def library_sha256():
    from mi355_paged import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()


def measured_traffic(config_name, lib_hash, batch):
    """(bytes per launch or None, note) from the committed counter digest of this config that was
    collected with the library now loaded, at the batch this run launches.  Every digest of the
    config is a candidate (file dates mean nothing after a checkout); the one whose library hash
    matches is taken, the highest name first among several."""
    import glob
    suffix = '_pmc_traffic.json' if config_name == 'metric' else f'_pmc_traffic_{config_name}.json'
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*' + suffix)), reverse=True)
    if not files:
        return None, {"traffic_stale": f"no profiles/*{suffix}"}
    seen = []
    for path in files:
        with open(path) as f:
            digest = json.load(f)
        rel = os.path.relpath(path, ROOT)
        if digest.get("library_sha256") != lib_hash:
            seen.append(f"{rel} ({str(digest.get('library_sha256'))[:12]})")
            continue
        # digests written before the batch was recorded were collected at the config's own batch
        digest_batch = digest.get("batch_per_gpu", CONFIGS[config_name]['B'])
        if digest_batch != batch:
            return None, {"traffic_stale": f"{rel} counted launches of {digest_batch} sequences, this run launches {batch}"}
        return digest.get("hbm_bytes_per_launch"), {"traffic_source": rel}
    return None, {"traffic_stale": "no digest was collected with the loaded build of the library "
                                   f"({lib_hash[:12]}); on file: " + ", ".join(seen)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='metric', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--fused-step', action='store_true',
                    help='cache write and attention in one call (mi355_paged_decode_append) instead of the two calls '
                         'of the reference step (mi355_kv_write_slots, mi355_paged_decode); named in config.workload')
    ap.add_argument('--strong', type=int, default=0, metavar='TOTAL',
                    help='strong scaling: TOTAL sequences over all ranks (TOTAL / N per GPU, e.g. 2048) instead of '
                         'the fixed per-GPU batch; reported as "scaling": "strong"')
    ap.add_argument('--gather-every', type=int, default=1,
                    help='steps whose sampled ids share one all-gather (N > 1; 1 = one collective per step, the '
                         'specified workload); every id is exchanged inside the timed region')
    args = ap.parse_args()
    # Keep stdout for the one JSON line: whatever libraries print on file descriptor 1 while the
    # job runs (the collective library prints a version banner there when its communicator comes
    # up) goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"bench.py --gpus {args.gpus}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device")
    # BENCH_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    device = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    import torch.distributed as dist
    # BENCH_FORCE_DIST=1 under a one-process torch.distributed.run: the process group, barrier,
    # asynchronous all-gather and max-reduce run on a one-rank RCCL communicator
    use_dist = world > 1 or bool(os.environ.get('BENCH_FORCE_DIST'))
    if use_dist:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    trace = (lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True)) \
        if os.environ.get('BENCH_TRACE') else (lambda m: None)
    if os.environ.get('BENCH_TRACE'):
        import faulthandler
        faulthandler.dump_traceback_later(40, exit=False, file=sys.stderr)
    trace('process group up')
    from mi355_paged import dp, ops
    cfg = dict(CONFIGS[args.config])
    if args.strong:
        if args.strong % world:
            sys.exit(f"bench.py --strong {args.strong}: not a multiple of {world} ranks")
        cfg['B'] = args.strong // world
    B, S, H, D, BS = cfg['B'], cfg['S'], cfg['H'], cfg['D'], cfg['BS']
    # small configs rotate through more pools so that a step never finds its KV in the 256 MiB Infinity Cache
    inp = build_inputs(cfg, device, seed=rank, sets=2 if args.config == 'metric' else 4)
    nsets = len(inp['pools'])
    out = torch.empty((B, H, D), dtype=torch.bfloat16, device=device)
    # How the ids are exchanged when there is more than one rank (BENCH_EXCHANGE overrides):
    #   device (default with RCCL): the library's own communicator (mi355_dp_*) on a side stream, each
    #          collective behind a one-wave watcher kernel that returns once the step's ids are written
    #          (mi355_dp_exchange_*): the compute stream carries the step's three kernels and nothing
    #          else, and no rank waits for another on it
    #   native: the same communicator behind an event record + stream wait per step
    #   torch : torch.distributed's all_gather_into_tensor on a side stream
    # A mode that cannot be set up on this machine falls back to the next one; the line says which ran.
    exchange = os.environ.get('BENCH_EXCHANGE') or ('native' if os.environ.get('BENCH_NATIVE_RCCL') else 'device')
    if not (use_dist and backend == 'nccl'):
        exchange = 'torch'
    if exchange == 'device' and args.gather_every != 1:
        exchange = 'native'
    exchange_note = None
    native = None
    if exchange in ('device', 'native'):
        try:
            native = dp.NativeComm(device)
        except Exception as e:   # no librccl.so.1, or the communicator did not come up
            exchange, exchange_note = 'torch', f"library communicator unavailable ({e}); torch.distributed instead"
    gather = dp.OverlappedTokenGather(B, device, single_rank_collective=use_dist, every=args.gather_every,
                                      native=native, device_started=exchange == 'device')

    roofline_ids = torch.empty((B,), dtype=torch.int32, device=device)
    ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(ROOFLINE_LAUNCHES)]
    ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(ROOFLINE_LAUNCHES)]

    # Cache write + attention of the step: one call (the wave that reads a sequence's last position
    # stores the new row there first), or the two calls it replaces.
    def attend(kc, vc, bt):
        if not args.fused_step:
            ops.paged_decode(inp['q'], kc, vc, bt, inp['ctx'], S, layout='NHSD', out=out)
        else:
            ops.paged_decode_append(inp['q'], inp['k_new'], inp['v_new'], kc, vc, bt, inp['ctx'], S, layout='NHSD',
                                    out=out)

    def step(i, bracket=None):
        kc, vc, bt = inp['pools'][i % nsets]
        if not args.fused_step:
            ops.kv_write_slots(kc, vc, inp['k_new'], inp['v_new'], inp['slots'][i % nsets], 'NHSD')
        if bracket is not None:
            ev_a[bracket].record()
        attend(kc, vc, bt)
        if bracket is not None:
            ev_b[bracket].record()
        if bracket is None:
            ops.sample_argmax(inp['logits'], out=gather.send_buffer(i))
            gather.launch(i)
        else:
            ops.sample_argmax(inp['logits'], out=roofline_ids)

    host_issue = [0.0]     # seconds the host spent issuing the last run's steps (before waiting for anything)

    def run_steps(first, count):
        h0 = time.perf_counter()
        for i in range(first, first + count):
            step(i)
        host_issue[0] = time.perf_counter() - h0
        gather.drain()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    trace('inputs built')
    # An idle card sits at its lowest clocks and needs tens of milliseconds of load to leave them
    # (a timing loop that starts cold reads up to 20 % slow for its first ~15 ms).  The step itself
    # runs for a quarter of a second first, outside the W warm-up steps and the timed region, so
    # that a short run (a few warm-up steps, 20 timed ones) measures the same card a long one does.
    # (A step count, not a deadline: every rank must start the same number of collectives.  About
    # 0.25 s of steps at 6 TB/s.)
    step_bytes = algorithmic_bytes(B, S, H, H, D, BS, 2, append=args.fused_step)
    clock_warm_steps = int(os.environ.get('BENCH_CLOCK_WARM_STEPS', min(10000, max(500, int(0.25 * 6e12 / step_bytes)))))
    first = 0
    while first < clock_warm_steps:
        n = min(200, clock_warm_steps - first)
        run_steps(first, n)
        first += n
    run_steps(first, args.warmup)
    first += args.warmup
    trace('warmup issued')
    barrier()
    trace('timed region')
    t0 = time.perf_counter()
    run_steps(first, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # every rank holds the whole batch's ids of the last step, its own slice among them
    last = first + args.steps - 1
    got = gather.result(last)
    if args.steps > 0:
        assert got.numel() == B * world
        if args.gather_every == 1 and not os.environ.get('MI355_DP_EXCHANGE_PROBE'):   # (probes drop parts of the exchange)
            mine = ops.sample_argmax(inp['logits'])      # (the logits are the same every step)
            assert torch.equal(got[rank * B:(rank + 1) * B], mine), "gathered ids do not hold this rank's own"

    # Roofline loop (outside `value`): the same steps, the decode launch of each between two HIP
    # events on the launch stream (torch's current stream, the one ops passes to the library).  A
    # spin kernel holds the stream for a few milliseconds first so that the host has queued the
    # launches before the device reaches them: an event pair then spans the device-side cost of the
    # launch (dispatch + kernel), not the host's call overhead, whatever the step's length.
    if rank == 0:
        torch.cuda._sleep(8_000_000)
        for j in range(ROOFLINE_LAUNCHES):
            step(first + args.steps + j, bracket=j)
        torch.cuda.synchronize()
        bracket_ms = sorted(ev_a[j].elapsed_time(ev_b[j]) for j in range(ROOFLINE_LAUNCHES))
        # A pair of events around ONE launch also spans the two event packets themselves (about 2.3 us
        # on this chip, whatever the kernel: the kernel trace of the same command is that much
        # shorter), which is 1.4 % of the metric launch and 9 % of a 24 us one.  The launch duration
        # the roofline uses therefore comes from ROOFLINE_GROUPS groups of ROOFLINE_GROUP decode
        # launches in a row (KV pools rotating, nothing else between them), one event pair around
        # each group: (kernel + dispatch gap) per launch, with the events' cost spread over the group.
        g_a = [torch.cuda.Event(enable_timing=True) for _ in range(ROOFLINE_GROUPS)]
        g_b = [torch.cuda.Event(enable_timing=True) for _ in range(ROOFLINE_GROUPS)]
        torch.cuda._sleep(8_000_000)
        for g in range(ROOFLINE_GROUPS):
            g_a[g].record()
            for j in range(ROOFLINE_GROUP):
                kc, vc, bt = inp['pools'][(g * ROOFLINE_GROUP + j) % nsets]
                attend(kc, vc, bt)
            g_b[g].record()
        torch.cuda.synchronize()
        kernel_ms = sorted(g_a[g].elapsed_time(g_b[g]) / ROOFLINE_GROUP for g in range(ROOFLINE_GROUPS))
        avg_ms = sum(kernel_ms) / len(kernel_ms)
    if use_dist:
        dist.barrier()
    abytes = algorithmic_bytes(B, S, H, H, D, BS, 2, append=args.fused_step)

    if rank == 0:
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        lib_hash = library_sha256()
        line = {
            "metric": "decode tokens/sec + achieved HBM GB/s, block_size=16 batch=256 seq=2048"
                      if args.config == 'metric' and not args.strong else
                      f"decode tokens/sec ({args.config}" + (f", {args.strong} sequences in total" if args.strong else "") + ")",
            "value": round(B * world * args.steps / elapsed, 1),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_warmup_steps": clock_warm_steps,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "host_issue_ms_per_step": round(1e3 * host_issue[0] / max(args.steps, 1), 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"paged decode step: batch={B}/GPU, ctx={S}, heads={H}, head_dim={D}, "
                                   f"block_size={BS}, bf16 KV [blocks,heads,{BS},{D}], scattered block table; "
                                   + ("kv-write and paged attention in one launch" if args.fused_step else
                                      "kv-write + paged attention")
                                   + f" + greedy argmax(vocab={cfg['vocab']})"
                                   + (" + RCCL all-gather of ids, one collective per step" if use_dist and args.gather_every == 1
                                      else f" + RCCL all-gather of ids, BATCHED: one collective per {args.gather_every} steps"
                                      if use_dist else ""),
                       "batch_per_gpu": B, "context": S, "block_size": BS, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "mi355_paged_decode_append (paged_decode_kernel, append mode)" if args.fused_step
                                   else "mi355_paged_decode (paged_decode_kernel + split merge)",
                         "algorithmic_bytes_per_launch": abytes,
                         "avg_launch_ms": round(avg_ms, 5),
                         "launches_timed": ROOFLINE_GROUPS * ROOFLINE_GROUP,
                         "p10_ms": round(kernel_ms[len(kernel_ms) // 10], 5),
                         "p50_ms": round(kernel_ms[len(kernel_ms) // 2], 5),
                         "p90_ms": round(kernel_ms[(len(kernel_ms) * 9) // 10], 5),
                         "timing": f"HIP events around groups of {ROOFLINE_GROUP} launches in a row (pools rotating), "
                                   f"{ROOFLINE_GROUPS} groups after the timed region; percentiles over groups",
                         "single_launch_brackets": {
                             "note": "one event pair per launch inside the same steps; includes the two event packets",
                             "launches": len(bracket_ms),
                             "avg_ms": round(sum(bracket_ms) / len(bracket_ms), 5),
                             "p10_ms": round(bracket_ms[len(bracket_ms) // 10], 5),
                             "p50_ms": round(bracket_ms[len(bracket_ms) // 2], 5),
                             "p90_ms": round(bracket_ms[(len(bracket_ms) * 9) // 10], 5),
                             "frac": round(abytes / (sum(bracket_ms) / len(bracket_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
            "library_sha256": lib_hash,
        }
        if use_dist:
            line["config"]["exchange"] = {
                "device": "library communicator (RCCL) on a side stream, started by the GPU: a one-wave watcher kernel "
                          "returns once the step's ids are written, then the all-gather runs (no event, no wait and "
                          "no extra launch on the compute stream)",
                "native": "library communicator (RCCL) on a side stream behind an event record + stream wait per step",
                "torch": "torch.distributed all_gather_into_tensor on a side stream"}[exchange]
            if exchange_note:
                line["config"]["exchange_note"] = exchange_note
        # HBM bytes per launch from the committed counter passes (tools/collect_profiles.sh +
        # tools/pmc_traffic.py; counters cannot be read inside this process)
        traffic, note = measured_traffic(args.config, lib_hash, B)
        if args.fused_step and traffic is not None:   # the counters were collected on the two-launch step's kernel
            traffic, note = None, {"traffic_stale": "counters were collected for mi355_paged_decode, not for the append kernel"}
        line["roofline"]["traffic"] = traffic
        line["roofline"].update(note)
        # the same achieved figure against what this box's HBM gives plain streaming kernels
        bw = measured_stream_bandwidth(device)
        line["roofline"]["measured_on_this_box"] = {
            "device_copy_1GiB_GBs": round(bw["copy"], 1), "read_only_sum_1GiB_GBs": round(bw["read"], 1),
            "achieved_over_device_copy": round(achieved / bw["copy"], 4)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, inp)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if native is not None:
        torch.cuda.synchronize()
        gather.close()
        native.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
