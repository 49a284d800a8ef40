#!/usr/bin/env python3
"""Whole-call time of the weight-only stem layer (3 -> 64, 7 x 7 / 2, 224 x 224, batch 256) per activation type:
HIP events, medians, outputs rotating over three buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B, dev = 256, torch.device("cuda", 0)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    x = torch.randn(B, 3, 224, 224, device=dev).to(dt)
    w = torch.randint(-128, 128, (64, 3, 7, 7), dtype=torch.int32, device=dev).to(torch.int8)
    alpha = torch.rand(64, device=dev) * 1e-3
    outs = [torch.empty(B, 64, 112, 112, device=dev) for _ in range(3)]
    it = [0]

    def call():
        N.conv2d_w8(x, w, 2, 3, alpha=alpha, out=outs[it[0] % 3])
        it[0] += 1
    print(f"{str(dt).split('.')[-1]:8s} stem: {bench.median_ms(call, 30, 10) * 1e3:7.1f} us", flush=True)
