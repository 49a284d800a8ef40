#!/usr/bin/env python3
"""One-rank RCCL self-test of the collective calls the sharded path makes (run under torchrun)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
world = dist.get_world_size()
M, n = 64, 32
y = torch.full((M, n), float(dist.get_rank() + 1), device=dev)
buf = torch.empty((world, M, n), device=dev)
work = dist.all_gather_into_tensor(buf.view(world * M, n), y, async_op=True)
work.wait()
torch.cuda.synchronize()
assert float(buf[0, 0, 0]) == 1.0
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("nccl selftest ok, world", world, flush=True)
dist.destroy_process_group()
