"""The multi-GPU path on CPU: two and three processes over gloo run the column-sharded layer
with a stand-in product (an exact integer matmul written here; the HIP GEMM needs a GPU) and
must reproduce the unsharded oracle result bit for bit, including a ragged split."""
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


def _worker(rank, world, port, n, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantization_amd.sharded import ColumnShardedLinear, run_stack, shard_rows
        g = torch.Generator().manual_seed(1)                # identical on every rank
        K, M = 48, 9
        w = torch.randn(n, K, generator=g)
        bias = torch.randn(n, generator=g)
        x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32).to(torch.int8)
        layer = ColumnShardedLinear.from_float(w, bias, gemm=_cpu_gemm, quantize=_cpu_quantize)
        assert (layer.lo, layer.hi) == shard_rows(n, world, rank)
        y = layer(x)
        w2 = torch.randn(n, K, generator=g)
        layer2 = ColumnShardedLinear.from_float(w2, None, gemm=_cpu_gemm, quantize=_cpu_quantize)
        ys = run_stack([layer, layer2], [x, x])
        if rank == 0:
            np.savez(tmp, y=y.numpy(), y0=ys[0].numpy(), y1=ys[1].numpy(), w=w.numpy(), w2=w2.numpy(),
                     bias=bias.numpy(), x=x.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (2, 37), (3, 64)])
def test_column_sharded_linear_equals_unsharded(tmp_path, world, n):
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    z = np.load(out)
    q, s, _ = O.quantize_channel_sym_int8(z["w"])
    ref = O.wo_linear_forward(z["x"], q, s, z["bias"])
    np.testing.assert_array_equal(z["y"], ref)
    np.testing.assert_array_equal(z["y0"], ref)
    q2, s2, _ = O.quantize_channel_sym_int8(z["w2"])
    np.testing.assert_array_equal(z["y1"], O.wo_linear_forward(z["x"], q2, s2, None))
