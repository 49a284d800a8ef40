#!/usr/bin/env python3
"""Tile A/B of the int8 GEMM at 4096^3 (one 256 x 256 tile per CU) through the developer switch MIQ_GEMM_TILE,
interleaved rounds in one process:  python tools/gemm_tiles_4096.py [size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = torch.randint(-128, 128, (S, S), dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (S, S), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(S, device="cuda") * 0.01
outs = [torch.empty(S, S, device="cuda") for _ in range(3)]
i = [0]


def step():
    N.gemm_i8(a, w, alpha=alpha, out=outs[i[0] % 3])
    i[0] += 1


def timed(iters=40):
    for _ in range(5):
        step()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        step()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for _ in range(300):
    step()
res = {}
for _ in range(5):
    for v in ("", "a", "d", "e", "m", "o", "p"):
        if v:
            os.environ["MIQ_GEMM_TILE"] = v
            os.environ["MIQ_GEMM_PLAN256"] = "0"
        else:
            os.environ.pop("MIQ_GEMM_TILE", None)
            os.environ.pop("MIQ_GEMM_PLAN256", None)
        res.setdefault(v, []).append(timed())
for v, t in res.items():
    m = sorted(t)[len(t) // 2]
    print(f"tile '{v or 'default 256x256'}': {m:7.1f} us  {2.0 * S ** 3 / m / 1e6:6.0f} TOPS", flush=True)
