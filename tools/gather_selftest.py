#!/usr/bin/env python3
"""Self-test of the collective calls the sharded path makes, under torchrun, with the backend of
MIQ_BENCH_BACKEND (nccl = RCCL, one GPU per rank; gloo = every rank on GPU 0, a rehearsal): the in-place
all-gather of ColumnShardedLinear on device tensors, a MAX all-reduce, a barrier.  Prints a line per step."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(40, exit=True)

backend = os.environ.get("MIQ_BENCH_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0"))
if backend != "nccl":
    local %= max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(backend)
world, rank = dist.get_world_size(), dist.get_rank()


def say(msg):
    if rank == 0:
        print(msg, flush=True)


say(f"{backend}: world {world}")
t = torch.tensor([1.5 + rank], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 0.5 + world
say("all_reduce ok")
dist.barrier()
say("barrier ok")
M, n = 64, 32
buf = torch.zeros((world, M * n), device=dev)
y = torch.full((M * n,), float(rank + 1), device=dev)
work = dist.all_gather_into_tensor(buf.view(-1), y, async_op=True)
work.wait()
torch.cuda.synchronize()
assert [float(buf[r, 0]) for r in range(world)] == [float(r + 1) for r in range(world)]
say("all_gather_into_tensor (separate input) ok")
buf.zero_()
buf[rank].fill_(float(rank + 1))
work = dist.all_gather_into_tensor(buf.view(-1), buf[rank], async_op=True)
work.wait()
torch.cuda.synchronize()
assert [float(buf[r, 5]) for r in range(world)] == [float(r + 1) for r in range(world)]
say("all_gather_into_tensor (in place) ok")
from quantization_amd.sharded import ColumnShardedLinear  # noqa: E402
g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(512, 256, generator=g, device=dev)
x = torch.randint(-128, 128, (32, 256), generator=g, dtype=torch.int32, device=dev).to(torch.int8)
for gd in (None, torch.bfloat16, torch.int8):
    layer = ColumnShardedLinear.from_float(w, None, gather_dtype=gd, act_scale=0.5 if gd == torch.int8 else None)
    y = layer(x)
    one = ColumnShardedLinear.__new__(ColumnShardedLinear)
    torch.cuda.synchronize()
    say(f"ColumnShardedLinear gather_dtype={gd}: {tuple(y.shape)} {y.dtype} ok")
# the bench's pattern at its rehearsal size: three layers in turn, each gathering its own buffer several times
for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32), ("i8", torch.int8)):
    big = torch.zeros((world, 2048 * 2048), dtype=dt, device=dev)
    for i in range(6):
        big[rank].fill_(rank + i)
        work = dist.all_gather_into_tensor(big.view(-1), big[rank], async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert [int(big[r, 7]) for r in range(world)] == [r + i for r in range(world)], (name, i)
    dist.barrier()
    t = torch.tensor([1.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    say(f"6 in-place gathers of [{world}, 2048*2048] {name} ok")
# the bench's step: a 2048^3 GEMM into the rank's slot, then the in-place gather, no device-wide wait in between
mode = os.environ.get("MIQ_SELFTEST_MODE", "inplace")
g = torch.Generator(device=dev).manual_seed(2)
w = torch.randn(2048, 2048, generator=g, device=dev)
x = torch.randint(-128, 128, (2048, 2048), generator=g, dtype=torch.int32, device=dev).to(torch.int8)
for gd in (torch.bfloat16, torch.float32):
    layer = ColumnShardedLinear.from_float(w, None, out_dtype=gd, out_features=2048 * world)
    for i in range(8):
        got = layer.compute_block(x, i & 1)
        if mode == "inplace":
            layer.start_gather(got).wait()
        else:
            src = got.buf[rank].clone()
            dist.all_gather_into_tensor(got.buf.view(-1), src, async_op=True).wait()
        print(f"rank {rank}: {gd} step {i} ({mode})", flush=True)
    torch.cuda.synchronize()
    dist.barrier()
dist.barrier()
say("selftest ok")
faulthandler.cancel_dump_traceback_later()
dist.destroy_process_group()
