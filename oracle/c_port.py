"""ctypes loader for oracle/paged_decode_ref.c (test / measurement infrastructure only)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liboracle_paged.so')
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle`")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_paged_decode_f32.restype = None
        _lib.oracle_paged_decode_f32.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6
    return _lib


def num_threads() -> int:
    return load().oracle_num_threads()


def paged_decode(query, k_cache, v_cache, block_tables, context_lens):
    """numpy float32 in -> numpy float32 out; query [B,Hq,D], caches [NB,Hkv,BS,D]."""
    q = np.ascontiguousarray(query, dtype=np.float32)
    k = np.ascontiguousarray(k_cache, dtype=np.float32)
    v = np.ascontiguousarray(v_cache, dtype=np.float32)
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    cl = np.ascontiguousarray(context_lens, dtype=np.int32)
    out = np.empty_like(q)
    B, Hq, D = q.shape
    load().oracle_paged_decode_f32(out.ctypes.data, q.ctypes.data, k.ctypes.data, v.ctypes.data,
                                   bt.ctypes.data, cl.ctypes.data, B, Hq, k.shape[1], D, k.shape[2],
                                   bt.shape[1])
    return out
