#!/usr/bin/env python3
"""Tile sweep of the im2col-order int8 convolution kernel (developer switch MIQ_CONV_TILE: a = 256x256, b = 256x128,
c = 256x64, d = 128x128, e = 128x64, 0 = the library's own choice) on the ResNet-18 layers that kernel serves, batch 256,
int32 out: whole calls, outputs rotating over three buffers, median of 5 interleaved rounds.

    python tools/conv_i8_tiles.py [variants, default 0abde]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

variants = sys.argv[1] if len(sys.argv) > 1 else "0abde"
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for C, H, Co, k, s, p in [(64, 56, 128, 3, 2, 1), (128, 28, 256, 3, 2, 1), (256, 14, 256, 3, 1, 1), (256, 14, 512, 3, 2, 1),
                          (512, 7, 512, 3, 1, 1), (256, 14, 512, 1, 2, 0)]:
    x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
    Ho = (H + 2 * p - k) // s + 1
    outs = [torch.empty(256, Co, Ho, Ho, dtype=torch.int32, device="cuda") for _ in range(3)]
    res = {}
    for _ in range(5):
        for v in variants:
            os.environ["MIQ_CONV_TILE"] = v
            for i in range(3):
                N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[i % 3])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[i % 3])
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{C}->{Co} {H}x{H} k{k}s{s}: " + " | ".join(f"{v} {statistics.median(t):6.1f}" for v, t in res.items()), flush=True)
