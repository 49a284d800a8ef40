#!/usr/bin/env python3
"""Per-layer timings of ResNet-18's convolutions at batch 256 as weight-only layers on floating-point activations
(miq_conv2d_w8, fp32 out): whole calls, HIP events, medians, outputs rotating over three buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B, dev = 256, torch.device("cuda", 0)
tot = {}
for dt in (torch.float32, torch.bfloat16, torch.float16):
    tot[dt] = 0.0
    for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
        x = torch.randn(B, C, H, H, device=dev).to(dt)
        w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
        alpha = torch.rand(Co, device=dev) * 1e-3
        Ho = (H + 2 * p - k) // s + 1
        outs = [torch.empty(B, Co, Ho, Ho, device=dev) for _ in range(3)]
        it = [0]

        def call():
            N.conv2d_w8(x, w, s, p, alpha=alpha, out=outs[it[0] % 3])
            it[0] += 1
        us = bench.median_ms(call, 10, 5) * 1e3
        tot[dt] += us * times
        fl = 2.0 * B * Co * Ho * Ho * C * k * k
        print(f"{str(dt).split('.')[-1]:8s} conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s} x{times}: {us:7.1f} us ({fl / us / 1e6:5.0f} TFLOP/s)", flush=True)
        del x, w, outs
    print(f"{str(dt).split('.')[-1]:8s} stack: {tot[dt]:.0f} us", flush=True)
