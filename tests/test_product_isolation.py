"""The product path has no CPU fallback and never touches the oracle (CPU only)."""
import os
import re
import subprocess
import sys

from conftest import PKG, ROOT


def _sources():
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                yield os.path.join(base, f)
    yield os.path.join(ROOT, 'include', 'mi355_paged.h')


def test_product_never_imports_the_oracle():
    for path in _sources():
        text = open(path).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), path
        assert 'paged_oracle' not in text and 'loop_form' not in text, path


def test_missing_library_fails_loudly():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mi355_paged import _lib\n"
            "try:\n    _lib.load()\nexcept RuntimeError as e:\n    print('RAISED', e)\n" % PKG)
    env = dict(os.environ, MI355_PAGED_LIB='/nonexistent/libmi355_paged.so')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert 'RAISED' in out.stdout and 'no CPU fallback' in out.stdout, out.stdout + out.stderr


def test_compute_entry_points_refuse_to_run_without_a_device():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    import mini_vllm
    import paged_attention as pa
    kv = torch.zeros(2, 2, 2, 16, 16)
    with pytest.raises(RuntimeError, match='HIP device'):
        pa.paged_attention_decode(torch.zeros(2, 16), kv, [0], 4, 16, 2, 2)
    with pytest.raises(RuntimeError, match='HIP device'):
        mini_vllm.Sampler.sample(torch.zeros(1, 8), [mini_vllm.Request('r', [1], 1, temperature=0.0)])
    eng = mini_vllm.MiniVLLM(vocab_size=50, hidden_size=32, num_layers=1, num_heads=2, intermediate_size=64,
                             num_kv_blocks=8, block_size=4)
    with pytest.raises(RuntimeError, match='HIP device'):
        eng.generate([[1, 2, 3]], max_new_tokens=2)
