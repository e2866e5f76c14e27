"""MI355X-native paged-attention decode + sampling kernels behind a C ABI.

`ops` holds the tensor-level wrappers, `_lib` the ctypes binding of
libmi355_paged.so, `dp` the data-parallel sharding helpers.
"""
from . import _lib, ops  # noqa: F401

__all__ = ['ops', '_lib']
