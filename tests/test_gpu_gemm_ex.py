"""miq_gemm_i8_ex through the C ABI: the transposed output, a column block of a wider matrix, and the int8 requantised
output of the int8 GEMM, each bit for bit against the C oracle (oracle.c: exact int32 product + the reference's
epilogue order) on every kernel family (weight-streaming, ring tiles, split-K, 256 x 256 pipelined, whole rounds +
tail, element-wise); then the column-sharded layer built on them, and a two-rank run (gloo, both ranks on this GPU)
of the real HIP GEMM writing into its slot of the in-place gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

SHAPES = [(1, 64, 64), (3, 4, 16), (16, 1024, 256), (17, 130, 48), (64, 256, 512), (129, 257, 400), (300, 4096, 256),
          (512, 768, 1024), (37, 130, 300), (16, 1024, 14336), (200, 128, 16384), (40, 520, 8192), (3584, 520, 64),
          (2304, 8192, 80), (520, 3584, 64), (4100, 24, 128), (24, 4100, 128)]


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def rand_i8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)


def operands(M, Nn, K):
    a, w = rand_i8((M, K), M * 7 + K), rand_i8((Nn, K), Nn * 13 + K)
    g = torch.Generator().manual_seed(M + Nn)
    alpha = torch.rand(Nn, generator=g) * 1e-3 + 1e-5
    bias = torch.randn(Nn, generator=g) * 100
    acc = OC.linear_i8_nt(a.numpy(), w.numpy()) if M * Nn * K <= 2 ** 28 else \
        O.linear_w8a8(a.numpy(), np.ascontiguousarray(w.numpy().T))
    return a, w, alpha, bias, acc


@pytest.mark.parametrize("M,Nn,K", SHAPES)
def test_transposed_output_bit_exact(N, M, Nn, K):
    a, w, alpha, bias, acc = operands(M, Nn, K)
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.25)
    ad, wd, al, bi = a.cuda(), w.cuda(), alpha.cuda(), bias.cuda()
    yt = N.gemm_i8(ad, wd, bias=bi, alpha=al, beta=0.25, transpose_out=True)
    assert tuple(yt.shape) == (Nn, M) and yt.is_contiguous()
    np.testing.assert_array_equal(yt.cpu().numpy(), ref.T)
    it = N.gemm_i8(ad, wd, out_dtype=torch.int32, transpose_out=True)
    np.testing.assert_array_equal(it.cpu().numpy(), acc.T)
    bt = N.gemm_i8(ad, wd, bias=bi, alpha=al, beta=0.25, out_dtype=torch.bfloat16, transpose_out=True)
    np.testing.assert_array_equal(bt.float().cpu().numpy(), O.round_f32_to_bf16(ref).T)
    # no bias; rows of a wider [N_total, M + pad] buffer (ldo > M)
    wide = torch.full((Nn, M + 12), -7.0, device="cuda")
    N.gemm_i8(ad, wd, alpha=al, beta=0.0, transpose_out=True, out=wide[:, 4:4 + M])
    np.testing.assert_array_equal(wide[:, 4:4 + M].cpu().numpy(), OC.epilogue_f32(acc, None, alpha.numpy(), 0.0).T)
    assert bool((wide[:, :4] == -7.0).all()) and bool((wide[:, 4 + M:] == -7.0).all())


@pytest.mark.parametrize("M,Nn,K", SHAPES)
@pytest.mark.parametrize("scale", [1.0, 37.5, 0.013])
def test_int8_requantised_output_bit_exact(N, M, Nn, K, scale):
    """out = int8(clamp(rint(y / scale))) with y the oracle's fp32 epilogue and numpy's IEEE division: the per-tensor
    form of Qop.quantize (quant/Weight_Only/Qop.py:154-157)."""
    a, w, alpha, bias, acc = operands(M, Nn, K)
    if scale != 37.5:
        alpha = alpha * (300.0 * scale)       # results that spread over the int8 range and saturate at both ends
    y = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), -0.5)
    ref = np.clip(np.rint(y / np.float32(scale)), -128, 127).astype(np.int8)
    ad, wd, al, bi = a.cuda(), w.cuda(), alpha.cuda(), bias.cuda()
    q = N.gemm_i8(ad, wd, bias=bi, alpha=al, beta=-0.5, out_dtype=torch.int8, requant_scale=scale)
    assert q.dtype == torch.int8
    np.testing.assert_array_equal(q.cpu().numpy(), ref)
    qt = N.gemm_i8(ad, wd, bias=bi, alpha=al, beta=-0.5, out_dtype=torch.int8, requant_scale=scale, transpose_out=True)
    np.testing.assert_array_equal(qt.cpu().numpy(), ref.T)


@pytest.mark.parametrize("M,Nn,K", [(9, 200, 320), (300, 1027, 512), (512, 768, 1024), (2304, 4096, 80), (16, 1024, 14336)])
def test_column_block_of_a_wider_matrix(N, M, Nn, K):
    """out[m * ldo + n]: a rank's columns written straight into a row-major [M, N_total] matrix."""
    a, w, alpha, bias, acc = operands(M, Nn, K)
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.0)
    total = Nn + 40
    for dt, conv in ((torch.float32, lambda r: r), (torch.bfloat16, O.round_f32_to_bf16)):
        full = torch.full((M, total), 3.0, dtype=dt, device="cuda")
        N.gemm_i8(a.cuda(), w.cuda(), bias=bias.cuda(), alpha=alpha.cuda(), out_dtype=dt, out=full[:, 24:24 + Nn])
        np.testing.assert_array_equal(full[:, 24:24 + Nn].float().cpu().numpy(), conv(ref))
        assert bool((full[:, :24] == 3.0).all()) and bool((full[:, 24 + Nn:] == 3.0).all())


def test_extended_entry_rejects_bad_arguments(N):
    a, w = rand_i8((8, 64), 1).cuda(), rand_i8((16, 64), 2).cuda()
    alpha = torch.ones(16, device="cuda")
    with pytest.raises(RuntimeError, match="requant_scale"):
        N.gemm_i8(a, w, alpha=alpha, out_dtype=torch.int8, requant_scale=0.0)
    with pytest.raises(RuntimeError, match="requant_scale"):
        N.gemm_i8(a, w, alpha=alpha, out_dtype=torch.int8, requant_scale=float("inf"))
    with pytest.raises(RuntimeError, match="wrong shape"):
        N.gemm_i8(a, w, alpha=alpha, transpose_out=True, out=torch.empty(8, 16, device="cuda"))
    with pytest.raises(RuntimeError, match="unit column stride"):
        N.gemm_i8(a, w, alpha=alpha, out=torch.empty(16, 8, device="cuda").t())


def test_sharded_layer_single_rank_forms(N):
    """ColumnShardedLinear on one GPU (the layer owns every row): the transposed placement and the int8 gather equal the
    plain call's result / its requantisation, results belong to the caller."""
    from quantization_amd.sharded import ColumnShardedLinear
    g = torch.Generator().manual_seed(4)
    w = (torch.randn(256, 320, generator=g) * 0.05).cuda()
    b = torch.randn(256, generator=g).cuda()
    q, scale, _ = N.quantize_rows_sym_i8(w)
    x1 = torch.randint(-128, 128, (48, 320), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    x2 = torch.randint(-128, 128, (48, 320), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    ref1 = N.gemm_i8(x1, q, bias=b, alpha=scale)
    plain = ColumnShardedLinear.from_float(w, b)
    y1 = plain(x1)
    plain(x2)
    assert torch.equal(y1, ref1)                      # not overwritten by the second forward
    tr = ColumnShardedLinear.from_float(w, b, transposed=True)
    t1 = tr(x1)
    tr(x2)
    assert torch.equal(t1, ref1) and t1.t().is_contiguous()
    acc = OC.linear_i8_nt(x1.cpu().numpy(), q.cpu().numpy())
    yref = OC.epilogue_f32(acc, b.cpu().numpy(), scale.cpu().numpy(), 0.0)
    np.testing.assert_array_equal(ref1.cpu().numpy(), yref)
    qi = ColumnShardedLinear.from_float(w, b, gather_dtype=torch.int8, act_scale=0.37)(x1)
    np.testing.assert_array_equal(qi.cpu().numpy(), np.clip(np.rint(yref / np.float32(0.37)), -128, 127).astype(np.int8))


# ---- two ranks sharing this GPU: the real GEMM into its slot of the in-place gather ----------------------------------

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantization_amd.sharded import ColumnShardedLinear, run_stack
        g = torch.Generator().manual_seed(1)                # identical on every rank
        n, K, M = 1000, 512, 300                            # a ragged split would need n % world != 0: see the second layer
        w, bias = torch.randn(n, K, generator=g) * 0.05, torch.randn(n, generator=g)
        w2 = torch.randn(n + 1, K, generator=g) * 0.05
        x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
        xd = x.cuda()
        out = {}
        for name, kw in (("f32", {}), ("bf16", {"gather_dtype": torch.bfloat16}), ("tr", {"transposed": True}),
                         ("i8", {"gather_dtype": torch.int8, "act_scale": 0.5}),
                         ("tr_i8", {"transposed": True, "gather_dtype": torch.int8, "act_scale": 0.5})):
            layer = ColumnShardedLinear.from_float(w.cuda(), bias.cuda(), **kw)
            y = layer(xd)
            again = layer.gather_async(xd, slot=0).to_dense()      # the cached, in-place buffer
            assert torch.equal(y, again)
            out[name] = y.float().cpu().numpy()
        ragged = ColumnShardedLinear.from_float(w2.cuda(), None)
        ragged_t = ColumnShardedLinear.from_float(w2.cuda(), None, transposed=True)
        ys = run_stack([ragged, ragged_t], [xd, xd])
        out["ragged"], out["ragged_t"] = ys[0].cpu().numpy(), ys[1].cpu().numpy()
        torch.cuda.synchronize()
        if rank == world - 1:
            np.savez(tmp, w=w.numpy(), w2=w2.numpy(), bias=bias.numpy(), x=x.numpy(), **out)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_real_gemm(tmp_path):
    """The N > 1 path with device tensors: every rank's HIP GEMM writes into its slice of the gather buffer and the
    all-gather runs in place on it (staged through the host for gloo).  Bit-equal to the unsharded oracle."""
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    q, s, _ = O.quantize_channel_sym_int8(z["w"])
    ref = O.wo_linear_forward(z["x"], q, s, z["bias"])
    np.testing.assert_array_equal(z["f32"], ref)
    np.testing.assert_array_equal(z["tr"], ref)
    np.testing.assert_array_equal(z["bf16"], O.round_f32_to_bf16(ref))
    qref = np.clip(np.rint(ref / np.float32(0.5)), -128, 127)
    np.testing.assert_array_equal(z["i8"], qref)
    np.testing.assert_array_equal(z["tr_i8"], qref)
    q2, s2, _ = O.quantize_channel_sym_int8(z["w2"])
    ref2 = O.wo_linear_forward(z["x"], q2, s2, None)
    np.testing.assert_array_equal(z["ragged"], ref2)
    np.testing.assert_array_equal(z["ragged_t"], ref2)


def test_split_k_repeated_on_two_streams_and_inside_a_graph(N):
    """Thin problems are cut along K into int32 slabs that a second launch sums: called repeatedly, on two streams, and
    recorded into a HIP graph that is replayed, the result stays the oracle's."""
    M, Nn, K = 40, 520, 8192
    a, w, alpha, bias, acc = operands(M, Nn, K)
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.0)
    ad, wd, al, bi = a.cuda(), w.cuda(), alpha.cuda(), bias.cuda()
    out = torch.empty(M, Nn, device="cuda")
    for _ in range(3):
        out.zero_()
        N.gemm_i8(ad, wd, bias=bi, alpha=al, out=out)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out2 = N.gemm_i8(ad, wd, bias=bi, alpha=al)
    out3 = N.gemm_i8(ad, wd, bias=bi, alpha=al)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out2.cpu().numpy(), ref)
    np.testing.assert_array_equal(out3.cpu().numpy(), ref)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        N.gemm_i8(ad, wd, bias=bi, alpha=al, out=out)
        N.gemm_i8(ad, wd, bias=bi, alpha=al, out=out)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
