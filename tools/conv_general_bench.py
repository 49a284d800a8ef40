#!/usr/bin/env python3
"""Timings of the direct convolution (miq_conv2d_general) on layers the MFMA kernels do not take, beside the library
convolution torch dispatches for the same arguments on a float copy of the weight: whole calls, HIP events, median of 20.

    python tools/conv_general_bench.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

# name, B, Cin, H, Cout, k, stride, padding, dilation, groups
LAYERS = [
    ("depthwise 3x3 (MobileNet 32ch 112^2)", 64, 32, 112, 32, 3, 1, 1, 1, 32),
    ("depthwise 3x3 s2 (144ch 56^2)", 64, 144, 56, 144, 3, 2, 1, 1, 144),
    ("depthwise 3x3 (384ch 14^2)", 64, 384, 14, 384, 3, 1, 1, 1, 384),
    ("grouped 3x3 g32 (ResNeXt 128ch 56^2)", 32, 128, 56, 128, 3, 1, 1, 1, 32),
    ("dilated 3x3 d2 (256ch 28^2)", 32, 256, 28, 256, 3, 1, 2, 2, 1),
    ("3x3 stride (2,1) (64ch 56^2)", 32, 64, 56, 64, 3, (2, 1), 1, 1, 1),
]
dev = torch.device("cuda", 0)
for name, B, C, H, Co, k, s, p, d, g in LAYERS:
    w = torch.randint(-128, 128, (Co, C // g, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    alpha = torch.rand(Co, device=dev) * 1e-3
    for xdt in (torch.int8, torch.float32, torch.bfloat16):
        x = (torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8) if xdt == torch.int8
             else torch.randn(B, C, H, H, device=dev, dtype=xdt))
        y = N.conv2d_general(x, w, s, p, d, g, alpha=alpha)
        t = bench.median_ms(lambda: N.conv2d_general(x, w, s, p, d, g, alpha=alpha, out=y), 20, 5) * 1e3
        ops = 2.0 * y.numel() * (C // g) * k * k
        byts = x.numel() * x.element_size() + y.numel() * 4 + w.numel()
        line = f"{name:40s} {str(xdt).split('.')[-1]:9s} {t:8.1f} us  {ops / t / 1e6:7.2f} T(FL)OPS  {byts / t / 1e3:7.1f} GB/s"
        if xdt != torch.int8:
            wf = w.to(xdt)
            tl = bench.median_ms(lambda: F.conv2d(x, wf, None, s, p, d, g), 20, 5) * 1e3
            line += f"   | library conv {tl:8.1f} us"
        print(line, flush=True)
