"""Engine-level data parallelism on the device: two ranks sharing the one GPU (gloo) must give
every request the tokens a single engine gives it on the whole batch, bit for bit under greedy
sampling.  The two-rank job is started by conftest.py once collection has selected this test -- child
processes are started before this process has made any GPU call -- and only its result is read here."""
import json
import os

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_engine_data_parallel_two_ranks_equal_single_engine(device):
    job = conftest.DP_JOB
    assert job is not None, "conftest did not start the two-rank job (needs a visible GPU)"
    proc, out_dir = job
    try:
        rc = proc.wait(timeout=240)
    except Exception:
        proc.kill()
        raise
    log = open(os.path.join(out_dir, 'job.log')).read()
    assert rc == 0, log[-3000:]
    want = json.load(open(os.path.join(out_dir, 'single.json')))
    assert len(want) == 9 and all(1 <= len(v) <= 8 for v in want.values())
    for r in range(2):
        got = json.load(open(os.path.join(out_dir, f'rank{r}.json')))
        assert got == want, f"rank {r} differs from the single-engine run"
