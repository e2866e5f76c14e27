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


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (MI355X)')


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
