"""Shared pytest configuration.

Markers: `gpu` = needs a real MI355X (run with `-m gpu`); everything else runs on CPU.
The product modules are imported by bare name from `vllm-from-scratch_amd/`, the same
way the reference's tests import `paged_attention` and `mini_vllm` from their own
directories.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'vllm-from-scratch_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


DP_JOB = None      # (Popen, output directory) of the two-rank engine job, when a GPU session started it


def _start_dp_job():
    """tests/dp_engine_job.py under torch.distributed.run, two ranks.  Started here, while this
    process has made no GPU call yet (counting devices does not initialise the runtime): a process
    that has initialised the GPU must not start other programs on this pool."""
    import socket
    import subprocess
    import tempfile
    out_dir = tempfile.mkdtemp(prefix='dp_engine_job_')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    log = open(os.path.join(out_dir, 'job.log'), 'w')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'dp_engine_job.py'), out_dir]
    return subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT), out_dir


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (MI355X)')


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """Starts the two-rank job only when its test is among the selected ones (after -m / -k have
    deselected), and before any test -- i.e. any GPU call -- has run in this process."""
    global DP_JOB
    wanted = any(item.fspath.basename == 'test_zy_dp_engine_gpu.py' for item in items)
    if wanted and DP_JOB is None:
        import torch
        if torch.cuda.device_count() > 0:
            DP_JOB = _start_dp_job()


def pytest_sessionfinish(session, exitstatus):
    """The job's processes do not outlive the session (a session that stopped before the test that
    waits for them, -x or an interrupt, would leave them behind)."""
    if DP_JOB is None:
        return
    proc = DP_JOB[0]
    if proc.poll() is None:
        proc.terminate()
        try:
            proc.wait(timeout=20)
        except Exception:
            proc.kill()
            proc.wait()


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def npz_cases(name):
    """Fixtures stored as c{i}_<field> arrays plus a JSON `meta` list."""
    z = load_npz(name)
    meta = json.loads(str(z['meta']))
    out = []
    for i, m in enumerate(meta):
        fields = {k[len(f'c{i}_'):]: z[k] for k in z.files if k.startswith(f'c{i}_')}
        out.append((m, fields))
    return out


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda', 0)
