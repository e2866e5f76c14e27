"""The C-ABI shared library: it loads without a GPU, exports every symbol the header
declares, and the ctypes binding names exactly those symbols (CPU only, no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from mi355_paged import _lib

HEADER = os.path.join(ROOT, 'include', 'mi355_paged.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mi355_[a-z0-9_]+)\s*\(', text)))


def test_library_present_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "build with make -C vllm-from-scratch_amd/csrc"
    lib = _lib.load()
    assert lib.mi355_abi_version() == _lib.ABI_VERSION == 1
    assert lib.mi355_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 9
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names


def test_host_only_entry_points():
    lib = _lib.load()
    # workspace sizing is pure host arithmetic
    assert lib.mi355_paged_decode_workspace_bytes(256, 8, 8, 64, 16, 2048, 1) % 16 == 0
    assert lib.mi355_paged_decode_workspace_bytes(1, 8, 8, 48, 16, 128, 1) == 0   # unsupported head_dim
    assert b'head_dim' in lib.mi355_last_error()
    # argument validation happens before any device work
    assert lib.mi355_paged_decode(None, None, 0, None, None, None, None, 1, 8, 8, 64, 16, 1, 1, 16,
                                  1024, 64, 64, 1.0, 1, None, 0, None) == -1
    assert lib.mi355_sample_argmax(None, None, 0, 10, 10, 0, None) == 0            # empty batch is a no-op


def test_ops_refuse_host_tensors():
    import torch
    from mi355_paged import ops
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.sample_argmax(torch.zeros(2, 8))
