"""Two-rank data-parallel engine run on one GPU (launched by tests/conftest.py under
torch.distributed.run before the test session touches the device; checked by
tests/test_zy_dp_engine_gpu.py).  Backend gloo: the ranks share the one GPU and the sampled ids
travel through host memory; the control flow is the one the RCCL run uses.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_engine_job.py <out_dir>

Every rank writes <out_dir>/rank<r>.json = the tokens of ALL requests as its front end returned
them; rank 0 also runs one engine on the whole batch and writes single.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vllm-from-scratch_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = dict(vocab_size=500, hidden_size=128, num_layers=2, num_heads=4, intermediate_size=256, num_kv_blocks=256,
           block_size=16, max_num_seqs=4, max_num_batched_tokens=64, enable_chunked_prefill=True, device='cuda')


def prompts():
    rng = np.random.default_rng(7)
    return [rng.integers(1, 500, size=int(n)).tolist() for n in (5, 70, 1, 33, 16, 120, 9, 48, 17)]


def main():
    out_dir = sys.argv[1]
    rank = int(os.environ['RANK'])
    dist.init_process_group('gloo')
    torch.cuda.set_device(0)
    import mini_vllm
    torch.manual_seed(1234 + rank)                  # ranks start from different weights; rank 0's are broadcast
    front = mini_vllm.DataParallelMiniVLLM(**CFG)
    got = front.generate(prompts(), max_new_tokens=8, temperature=0.0)
    with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as f:
        json.dump(got, f)
    if rank == 0:
        single = mini_vllm.MiniVLLM(**CFG)
        state = {k: v for k, v in front.engine.model.state_dict().items() if k not in ('k_cache', 'v_cache')}
        single.model.load_state_dict(state, strict=False)
        ids = [single.add_request(p, max_new_tokens=8, temperature=0.0) for p in prompts()]
        want = {rid: [] for rid in ids}
        while single.scheduler.has_unfinished:
            for rid, tok in single.step():
                want[rid].append(tok)
        with open(os.path.join(out_dir, 'single.json'), 'w') as f:
            json.dump(want, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
