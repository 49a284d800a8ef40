"""miq_gemm_i8_noalloc (the entry point that allocates nothing) against miq_gemm_i8, through the C ABI.

Two calls in a row on one GPU, the second on a smaller problem, each compared bit for bit with the existing
entry point run on the same operands with a caller workspace.  The shapes are chosen so that both kinds of
scratch are exercised: the int32 slabs of a split-K run (thin problem, [N,K] weight) and the re-laid copy of a
[K,N] weight.  Collected last (file name), because the round's end runs `pytest -x`.
"""
import ctypes

import pytest
import torch

from quantization_amd import _native

F32, BF16, I32 = _native.MIQ_F32, _native.MIQ_BF16, _native.MIQ_I32
NK, KN = _native.MIQ_W_NK, _native.MIQ_W_KN
_p = ctypes.c_void_p


def _ptr(t):
    return _p(t.data_ptr()) if t is not None else _p(0)


def _operands(M, N, K, layout, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (N, K) if layout == NK else (K, N), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(N, generator=g, device="cuda") * 1e-3 + 1e-5
    bias = torch.randn(N, generator=g, device="cuda") * 50
    return a, w, alpha, bias


def _both(lib, M, N, K, layout, out_code, seed):
    """(noalloc result, miq_gemm_i8 result, bytes of scratch the problem needs)."""
    a, w, alpha, bias = _operands(M, N, K, layout, seed)
    dt = {F32: torch.float32, BF16: torch.bfloat16, I32: torch.int32}[out_code]
    stream = _p(torch.cuda.current_stream().cuda_stream)
    need = lib.miq_gemm_i8_workspace(layout, M, N, K)
    # poisoned outputs: a kernel that wrote nothing cannot pass
    got = torch.full((M, N), -7, dtype=dt, device="cuda")
    ref = torch.full((M, N), -9, dtype=dt, device="cuda")
    rc = lib.miq_gemm_i8_noalloc(_ptr(a), _ptr(w), layout, M, N, K, _ptr(bias), _ptr(alpha), 0.5, _ptr(got), out_code, stream)
    assert rc == 0, lib.miq_last_error().decode()
    ws = torch.empty(max(need, 256), dtype=torch.int8, device="cuda")
    rc = lib.miq_gemm_i8(_ptr(a), _ptr(w), layout, M, N, K, _ptr(bias), _ptr(alpha), 0.5, _ptr(ref), out_code, _ptr(ws),
                         ws.numel(), stream)
    assert rc == 0, lib.miq_last_error().decode()
    torch.cuda.synchronize()
    return got, ref, need


@pytest.mark.gpu
def test_noalloc_matches_gemm_i8_on_two_calls_in_a_row():
    lib = _native.lib()
    limit = lib.miq_gemm_i8_noalloc_max_workspace()
    assert limit == 32 << 20
    # first call: split-K slabs ([N,K] weight, thin problem); second, smaller: slabs again, fewer of them
    for M, N, K, layout, code, seed in ((200, 512, 16384, NK, F32, 1), (40, 256, 8192, NK, F32, 2)):
        got, ref, need = _both(lib, M, N, K, layout, code, seed)
        assert 0 < need <= limit, "the shape was chosen to need split-K slabs"
        assert torch.equal(got, ref)
    # a [K,N] weight (re-laid into the scratch) with split-K slabs behind it, then a smaller one without slabs,
    # in the other output types
    for M, N, K, layout, code, seed in ((64, 768, 8192, KN, I32, 3), (300, 640, 512, KN, BF16, 4), (8, 64, 64, NK, F32, 5)):
        got, ref, need = _both(lib, M, N, K, layout, code, seed)
        assert need <= limit
        assert torch.equal(got, ref)
    # the larger problem once more after the smaller ones: stale scratch contents must not matter
    got, ref, _ = _both(lib, 200, 512, 16384, NK, F32, 1)
    assert torch.equal(got, ref)


@pytest.mark.gpu
def test_noalloc_rejects_what_does_not_fit():
    lib = _native.lib()
    a, w, alpha, bias = _operands(16, 8192, 8192, KN, 6)     # the re-laid weight alone is 64 MiB
    out = torch.full((16, 8192), 3.0, device="cuda")
    rc = lib.miq_gemm_i8_noalloc(_ptr(a), _ptr(w), KN, 16, 8192, 8192, _ptr(bias), _ptr(alpha), 0.0, _ptr(out), F32,
                                 _p(torch.cuda.current_stream().cuda_stream))
    assert rc == -3 and b"scratch" in lib.miq_last_error()
    torch.cuda.synchronize()
    assert bool((out == 3.0).all()), "a rejected call must not touch the output"
