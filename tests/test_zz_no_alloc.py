"""mi355_paged_decode_static: the decode entry point that allocates nothing (needs an MI355X).

Called through the C ABI (ctypes) with tensors allocated beforehand and compared with
mi355_paged_decode on the same inputs.  Both run the same kernels with the same launch plan, the
only difference being where the split-KV partials live, so the bar is bit equality; the result
is also held to the oracle at the usual tolerance.  Two calls in a row share the built-in
scratch array, the second on a smaller input (stale records of the first call stay in the array
and must not be read).
"""
import math

import numpy as np
import pytest
import torch

from oracle import paged_oracle as orc

pytestmark = pytest.mark.gpu

LOWP_TOL = dict(atol=1e-2, rtol=1.6e-2)
F32_TOL = dict(atol=1e-5, rtol=1e-5)
_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _problem(rng, device, dtype, B, Hq, Hkv, D, BS, ctx):
    per = (max(ctx) + BS - 1) // BS
    NB = B * per + 5
    cache = torch.from_numpy(rng.standard_normal((2, NB, Hkv, BS, D), dtype=np.float32)).to(dtype)
    q = torch.from_numpy(rng.standard_normal((B, Hq, D), dtype=np.float32)).to(dtype)
    bt = rng.permutation(NB)[:B * per].reshape(B, per).astype(np.int32)
    want = orc.paged_decode_batch(q.float().numpy(), cache[0].float().numpy(), cache[1].float().numpy(), bt,
                                  ctx, BS, dtype=np.float64)
    return dict(q=q.to(device), k=cache[0].to(device), v=cache[1].to(device),
                bt=torch.from_numpy(bt).to(device), ctx=torch.tensor(ctx, dtype=torch.int32, device=device),
                B=B, Hq=Hq, Hkv=Hkv, D=D, BS=BS, per=per, NB=NB, max_ctx=max(ctx), want=want, dtype=dtype)


def _args(pr, out):
    k = pr['k']
    return (out.data_ptr(), pr['q'].data_ptr(), pr['q'].stride(0), k.data_ptr(), pr['v'].data_ptr(),
            pr['bt'].data_ptr(), pr['ctx'].data_ptr(), pr['B'], pr['Hq'], pr['Hkv'], pr['D'], pr['BS'], pr['per'],
            pr['NB'], pr['max_ctx'], k.stride(0), k.stride(1), k.stride(2), 1.0 / math.sqrt(pr['D']),
            _CODE[pr['dtype']])


def _both(lib, pr, device, stream):
    """(result of mi355_paged_decode with a caller workspace, result of mi355_paged_decode_static,
    workspace bytes the shape needs); every tensor is allocated before the static call."""
    need = lib.mi355_paged_decode_workspace_bytes(pr['B'], pr['Hq'], pr['Hkv'], pr['D'], pr['BS'], pr['max_ctx'],
                                                  _CODE[pr['dtype']])
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
    ref = torch.full((pr['B'], pr['Hq'], pr['D']), float('nan'), dtype=pr['dtype'], device=device)
    got = torch.full_like(ref, float('nan'))
    rc = lib.mi355_paged_decode(*_args(pr, ref), ws.data_ptr() if need else None, need, stream)
    assert rc == 0, lib.mi355_last_error()
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats(device)['allocation.all.allocated']
    rc = lib.mi355_paged_decode_static(*_args(pr, got), stream)
    assert rc == 0, lib.mi355_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats(device)['allocation.all.allocated'] == before   # torch's allocator untouched
    return ref, got, need


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_static_scratch_decode_equals_workspace_decode_twice(device, dtype):
    from mi355_paged import _lib
    lib = _lib.load()
    assert lib.mi355_abi_version() == _lib.ABI_VERSION
    rng = np.random.default_rng(20 + _CODE[dtype])
    tol = F32_TOL if dtype == torch.float32 else LOWP_TOL
    stream = torch.cuda.current_stream(device).cuda_stream
    # first call: few long sequences -> the context is split across workgroups (partials + merge kernel);
    # second call: fewer, shorter sequences -> fewer splits, the array still holds the first call's records
    shapes = [dict(B=3, Hq=8, Hkv=8, D=64, BS=16, ctx=[6000, 8192, 37]),
              dict(B=2, Hq=8, Hkv=8, D=64, BS=16, ctx=[2100, 900])]
    needs = []
    with torch.cuda.device(device):
        for sh in shapes:
            pr = _problem(rng, device, dtype, **sh)
            ref, got, need = _both(lib, pr, device, stream)
            needs.append(need)
            assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))           # bit for bit
            np.testing.assert_allclose(got.float().cpu().numpy(), pr['want'], **tol)
    assert needs[0] > 0 and needs[1] > 0 and needs[1] < needs[0], needs   # both calls used the scratch array


def test_static_scratch_grouped_query_and_unsplit_shapes(device):
    """The matrix-core kernel for grouped-query shapes shares the partials format; a shape that
    needs no partials never touches the array."""
    from mi355_paged import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        pr = _problem(rng, device, torch.bfloat16, B=2, Hq=32, Hkv=4, D=128, BS=16, ctx=[5000, 4096])
        ref, got, need = _both(lib, pr, device, stream)
        assert need > 0
        assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
        np.testing.assert_allclose(got.float().cpu().numpy(), pr['want'], **LOWP_TOL)
        pr = _problem(rng, device, torch.bfloat16, B=300, Hq=8, Hkv=8, D=64, BS=16, ctx=[64] * 300)
        ref, got, need = _both(lib, pr, device, stream)
        assert need == 0
        assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
        np.testing.assert_allclose(got.float().cpu().numpy(), pr['want'], **LOWP_TOL)


def test_static_scratch_rejects_shapes_beyond_the_array(device):
    from mi355_paged import _lib
    lib = _lib.load()
    limit = 16 << 20
    # 64 sequences x 64 heads x D 256 over a million tokens each: a workgroup stages at most 8190 block-table
    # entries, so every sequence is cut into 8 splits whatever the batch, and their partials exceed the array
    B, Hq, Hkv, D, BS, ctx = 64, 64, 64, 256, 16, 1 << 20
    need = lib.mi355_paged_decode_workspace_bytes(B, Hq, Hkv, D, BS, ctx, 1)
    assert need > limit, need
    dummy = torch.zeros(64, dtype=torch.uint8, device=device)
    p = dummy.data_ptr()
    with torch.cuda.device(device):
        rc = lib.mi355_paged_decode_static(p, p, Hq * D, p, p, p, p, B, Hq, Hkv, D, BS, ctx // BS, 1 << 20, ctx,
                                           Hkv * BS * D, BS * D, D, 0.0625, 1, None)
    assert rc == -3
    msg = lib.mi355_last_error().decode()
    assert 'built-in scratch' in msg and str(need) in msg, msg
    torch.cuda.synchronize()


def test_static_scratch_decode_under_graph_capture(device):
    """The call is made for code under HIP graph capture: captured (after one eager call on the device,
    which looks the scratch array's address up) and replayed on changed inputs, a split shape gives
    what mi355_paged_decode gives on those inputs; the driver's free-memory figure does not move
    around the eager call nor around the replays (a hipMalloc inside the library would show there,
    not in torch's allocator statistics)."""
    from mi355_paged import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    with torch.cuda.device(device):
        pr = _problem(rng, device, torch.bfloat16, B=3, Hq=8, Hkv=8, D=64, BS=16, ctx=[6000, 8192, 37])
        need = lib.mi355_paged_decode_workspace_bytes(pr['B'], pr['Hq'], pr['Hkv'], pr['D'], pr['BS'], pr['max_ctx'], 1)
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        ref = torch.empty((pr['B'], pr['Hq'], pr['D']), dtype=torch.bfloat16, device=device)
        got = torch.full_like(ref, float('nan'))
        q2 = (pr['q'].float() * -0.5 + 0.25).to(torch.bfloat16)          # the replay's inputs, allocated beforehand
        ctx2 = torch.tensor([8000, 100, 8192], dtype=torch.int32, device=device)
        side = torch.cuda.Stream(device=device)
        with torch.cuda.stream(side):
            # the eager call a device needs before captures: the library's device image (its code and the
            # scratch array with it) is loaded by the first call, and the array's address is looked up
            rc = lib.mi355_paged_decode_static(*_args(pr, got), side.cuda_stream)
            assert rc == 0, lib.mi355_last_error()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(device)[0]
        with torch.cuda.stream(side):
            rc = lib.mi355_paged_decode_static(*_args(pr, got), side.cuda_stream)
            assert rc == 0, lib.mi355_last_error()
        side.synchronize()
        assert torch.cuda.mem_get_info(device)[0] == free0          # the call took no device memory
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            rc = lib.mi355_paged_decode_static(*_args(pr, got), torch.cuda.current_stream(device).cuda_stream)
            assert rc == 0, lib.mi355_last_error()
        inputs = ((pr['q'].clone(), pr['ctx'].clone()), (q2, ctx2))
        outs = [torch.empty_like(got) for _ in inputs]
        graph.replay()                                              # (the first replay uploads the graph)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(device)[0]                  # (capturing is the framework's business)
        for (q_new, ctx_new), keep in zip(inputs, outs):
            pr['q'].copy_(q_new)
            pr['ctx'].copy_(ctx_new)
            graph.replay()
            keep.copy_(got)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(device)[0] == free0
        for (q_new, ctx_new), keep in zip(inputs, outs):
            pr['q'].copy_(q_new)
            pr['ctx'].copy_(ctx_new)
            rc = lib.mi355_paged_decode(*_args(pr, ref), ws.data_ptr(), need, torch.cuda.current_stream(device).cuda_stream)
            assert rc == 0, lib.mi355_last_error()
            torch.cuda.synchronize()
            assert torch.equal(keep.view(torch.uint8), ref.view(torch.uint8))
