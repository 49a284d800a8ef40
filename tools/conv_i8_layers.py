#!/usr/bin/env python3
"""ResNet-18's eleven distinct int8 convolutions at batch 256 (int8 NCHW in, int32 NCHW out, filters prepacked), whole call,
each timed as a replayed HIP graph of six calls with the output rotating over three buffers, and the stack's total.
For an A/B of two builds on one box run it twice in one GPU call:

    [MIQ_LIB=path/to/libmiq.so] python tools/conv_i8_layers.py [filter: e.g. 128,28]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

if os.environ.get("MIQ_LIB"):
    N.LIB_PATH = os.path.abspath(os.environ["MIQ_LIB"])
only = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else None
dev = torch.device("cuda", 0)
B = 256
total = 0.0
for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
    if only and (C, H) != only[:2]:
        continue
    x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    Ho = (H + 2 * p - k) // s + 1
    outs = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
    pk = N.conv2d_i8_prepack(w, (B, C, H, H), s, p)
    it = [0]

    def step():
        N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[it[0] % 3], packed=pk)
        it[0] += 1
    for _ in range(40):
        step()
    ms = sorted(bench.graphed_ms(step, 6, 10, 6) for _ in range(3))[1]
    total += ms * times
    print(f"conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s} x{times}: {ms * 1e3:7.1f} us", flush=True)
    del x, w, outs
print(f"stack: {total * 1e3:.0f} us", flush=True)
