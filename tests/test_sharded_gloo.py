"""The multi-GPU path on CPU: 2, 3, 4 and 8 processes over gloo run the column-sharded layer with a
stand-in product (an exact integer matmul written here; the HIP GEMM needs a GPU) and must reproduce
the unsharded oracle result bit for bit, including ragged splits, the Llama-3-70B width (28672 output
channels), ranks that were handed only their own rows, and the bf16 / int8 forms of the gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import oracle_np as O

ACT_SCALE = 37.5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_gemm(x, w, bias=None, alpha=None, beta=0.0, out_dtype=torch.float32):
    acc = (x.to(torch.int64) @ w.to(torch.int64).T).to(torch.int32)
    if out_dtype == torch.int32:
        return acc
    y = acc.to(torch.float32)
    if bias is not None:
        y = y + bias
    return (y * alpha + beta).to(out_dtype)


def _cpu_quantize(w):
    q, s, _ = O.quantize_channel_sym_int8(w.numpy())
    return torch.from_numpy(q), torch.from_numpy(s.reshape(-1))


def _worker(rank, world, port, n, K, M, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantization_amd.sharded import ColumnShardedLinear, run_stack, shard_rows
        g = torch.Generator().manual_seed(1)                # identical on every rank
        w = torch.randn(n, K, generator=g)
        bias = torch.randn(n, generator=g)
        x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
        layer = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize)
        assert (layer.lo, layer.hi) == shard_rows(n, world, rank)
        y = layer(x)
        y_again = layer(x)                                   # the cached gather buffer, reused
        blocks = layer(x, dense=False)
        assert [tuple(b.shape) for b in blocks] == [(M, shard_rows(n, world, r)[1] - shard_rows(n, world, r)[0])
                                                    for r in range(world)]
        assert torch.equal(torch.cat(blocks, dim=1), y) and torch.equal(y_again, y)
        # a rank that was handed only its own rows (how large models are loaded)
        lo, hi = shard_rows(n, world, rank)
        own = ColumnShardedLinear.from_float(w[lo:hi].clone(), bias[lo:hi].clone(), gemm=_cpu_gemm, quantize=_cpu_quantize,
                                             out_features=n)
        assert torch.equal(own(x), y)
        w2 = torch.randn(n, K, generator=g)
        layer2 = ColumnShardedLinear.from_float(w2, None, gemm=_cpu_gemm, quantize=_cpu_quantize)
        ys = run_stack([layer, layer2], [x, x])
        yb = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize, gather_dtype=torch.bfloat16)(x)
        yq = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize, gather_dtype=torch.int8,
                                            act_scale=ACT_SCALE)(x)
        # the transposed placement: every rank gathers out^T; to_dense() is a view of the gathered buffer when N divides
        # by the world size
        lt = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize, transposed=True)
        yt = lt(x)
        assert tuple(yt.shape) == (M, n)
        if n % world == 0:
            assert yt.t().is_contiguous()                    # [N, M] storage: no re-arranging pass happened
        assert torch.equal(torch.cat(lt(x, dense=False), dim=1), yt)
        ytq = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize, transposed=True,
                                             gather_dtype=torch.int8, act_scale=ACT_SCALE)(x)
        # results are the caller's: a second forward on other activations leaves the first result alone
        x2 = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
        keep = y.clone()
        blocks_keep = [b.clone() for b in blocks]
        layer(x2)
        layer(x2, dense=False)
        assert torch.equal(y, keep) and all(torch.equal(a, b) for a, b in zip(blocks, blocks_keep))
        if rank == world - 1:                                # the last rank: the narrowest shard of a ragged split
            np.savez(tmp, y=y.numpy(), y0=ys[0].numpy(), y1=ys[1].numpy(), w=w.numpy(), w2=w2.numpy(),
                     bias=bias.numpy(), x=x.numpy(), yb=yb.float().numpy(), yq=yq.numpy(), yt=yt.numpy(), ytq=ytq.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,K,M", [(2, 64, 48, 9), (2, 37, 48, 9), (3, 64, 48, 9), (3, 28672, 32, 5),
                                         (4, 28672, 32, 5), (8, 28672, 32, 5), (8, 8192, 64, 3)])
def test_column_sharded_linear_equals_unsharded(tmp_path, world, n, K, M):
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, K, M, out), nprocs=world, join=True)
    z = np.load(out)
    q, s, _ = O.quantize_channel_sym_int8(z["w"])
    ref = O.wo_linear_forward(z["x"], q, s, z["bias"])
    np.testing.assert_array_equal(z["y"], ref)
    np.testing.assert_array_equal(z["y0"], ref)
    q2, s2, _ = O.quantize_channel_sym_int8(z["w2"])
    np.testing.assert_array_equal(z["y1"], O.wo_linear_forward(z["x"], q2, s2, None))
    # what the narrower gathers deliver: the fp32 result rounded once to bf16 / requantised with the static scale
    np.testing.assert_array_equal(z["yb"], O.round_f32_to_bf16(ref))
    np.testing.assert_array_equal(z["yq"], np.clip(np.rint(ref / np.float32(ACT_SCALE)), -128, 127).astype(np.int8))
    np.testing.assert_array_equal(z["yt"], ref)
    np.testing.assert_array_equal(z["ytq"], z["yq"])


def test_single_rank_results_are_owned_by_the_caller():
    """One rank (no process group): y1 = layer(x1); y2 = layer(x2) must leave y1 alone, for the dense result, the block
    list and the transposed placement alike (a cached gather buffer handed out as the result would be overwritten)."""
    from quantization_amd.sharded import ColumnShardedLinear
    g = torch.Generator().manual_seed(3)
    w, bias = torch.randn(24, 16, generator=g), torch.randn(24, generator=g)
    x1 = torch.randint(-128, 128, (5, 16), generator=g, dtype=torch.int32).to(torch.int8)
    x2 = torch.randint(-128, 128, (5, 16), generator=g, dtype=torch.int32).to(torch.int8)
    q, s, _ = O.quantize_channel_sym_int8(w.numpy())
    ref1 = O.wo_linear_forward(x1.numpy(), q, s, bias.numpy())
    for transposed in (False, True):
        layer = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize, transposed=transposed)
        y1 = layer(x1)
        b1 = layer(x1, dense=False)
        y2 = layer(x2)
        layer(x2, dense=False)
        np.testing.assert_array_equal(y1.numpy(), ref1)
        np.testing.assert_array_equal(b1[0].numpy(), ref1)
        assert not torch.equal(y1, y2)
        # the cached form says so in its contract: same slot, same rows -> the buffer is reused
        a = layer.gather_async(x1, slot=0).to_dense()
        layer.gather_async(x2, slot=0)
        np.testing.assert_array_equal(a.numpy(), y2.numpy())


def test_from_float_checks_the_rows_it_is_given():
    from quantization_amd.sharded import ColumnShardedLinear
    w = torch.randn(10, 8)
    with pytest.raises(RuntimeError, match="expected 12 rows"):
        ColumnShardedLinear.from_float(w, None, gemm=_cpu_gemm, quantize=_cpu_quantize, out_features=12)
    with pytest.raises(RuntimeError, match="act_scale"):
        ColumnShardedLinear.from_float(w, None, gemm=_cpu_gemm, quantize=_cpu_quantize, gather_dtype=torch.int8)
