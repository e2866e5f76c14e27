"""ctypes binding of libmi355_paged.so (the C ABI declared in include/mi355_paged.h).

There is no fallback: if the shared library is missing or cannot be loaded the
first use raises, and every compute entry point requires a HIP device.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_PAGED_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get('MI355_PAGED_LIB') or os.path.join(_HERE, 'libmi355_paged.so')
ABI_VERSION = 1

# name -> (restype, argtypes); mirrors include/mi355_paged.h one to one
SIGNATURES = {
    'mi355_abi_version': (c_int, []),
    'mi355_last_error': (c_char_p, []),
    'mi355_paged_decode': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    'mi355_paged_decode_append': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    'mi355_paged_decode_append_rope': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
        c_void_p, c_void_p, c_int32,
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    'mi355_paged_decode_static': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_float, c_int32, c_void_p]),
    'mi355_paged_decode_workspace_bytes': (c_size_t, [
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    'mi355_kv_write_slots': (c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
        c_int64, c_int32, c_int32, c_int64, c_int64, c_int64, c_int32, c_void_p]),
    'mi355_prefill_causal': (c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int32,
        c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_void_p]),
    'mi355_sample_argmax': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_void_p]),
    'mi355_sample_topk_topp': (c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_int32, c_int32,
        c_int64, c_int32, c_void_p, c_void_p]),
    'mi355_mla_decode': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
        c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    'mi355_mla_decode_paged': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
        c_void_p, c_int32, c_int32, c_int32,
        c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_float, c_int32, c_void_p, c_size_t, c_void_p]),
    'mi355_mla_decode_workspace_bytes': (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    'mi355_slot_mapping': (c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'mi355_paged_prefill': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
        c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_float, c_int32, c_void_p]),
    'mi355_paged_prefill_max_context': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'mi355_decode_step_inputs': (c_int, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
        c_int32, c_int32, c_int32, c_void_p]),
    'mi355_rope_kv_write': (c_int, [
        c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int32, c_void_p,
        c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
        c_int64, c_int64, c_int64, c_int32, c_void_p]),
    'mi355_rmsnorm': (c_int, [
        c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_float, c_int32, c_int32,
        c_int32, c_void_p]),
    'mi355_dp_unique_id': (c_int, [c_void_p]),
    'mi355_dp_init': (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    'mi355_dp_allgather_i32': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'mi355_dp_destroy': (c_int, [c_void_p]),
    'mi355_dp_exchange_create': (c_int, [c_void_p, c_void_p]),
    'mi355_dp_exchange_arm': (c_int, [c_void_p, c_void_p, c_int64]),
    'mi355_dp_exchange_launch': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
    'mi355_dp_exchange_launch_after': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
    'mi355_dp_exchange_done': (c_int, [c_void_p, c_int32]),
    'mi355_dp_exchange_wait': (c_int, [c_void_p, c_int32, c_void_p, c_int32]),
    'mi355_dp_exchange_destroy': (c_int, [c_void_p]),
    'mi355_debug_set_knob': (c_int, [c_char_p, c_int32]),
    'mi355_silu_mul': (c_int, [
        c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
}

_lib = None


class Mi355Error(RuntimeError):
    """A library call returned a non-zero status."""


def load() -> ctypes.CDLL:
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"mi355_paged: {LIB_PATH} not found. Build it with "
            f"`make -C vllm-from-scratch_amd/csrc` (or __graft_entry__.build()); "
            f"there is no CPU fallback for this path.")
    import torch  # noqa: F401  (loads the HIP runtime this library binds to)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = lib.mi355_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"mi355_paged: ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().mi355_last_error().decode('utf-8', 'replace')
        raise Mi355Error(f"{what} failed (status {status}): {msg}")
