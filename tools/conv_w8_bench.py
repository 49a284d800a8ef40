#!/usr/bin/env python3
"""Weight-only Conv2d on floating-point activations: miq_conv2d_w8 (whole call: re-layout + implicit GEMM) beside the
library route it replaces (float convolution on a cached float copy of the weight + the fused epilogue kernel),
interleaved in one process, ResNet-18 shapes at batch 256 and the reference's sample-model shapes.

    python tools/conv_w8_bench.py [rounds] [iters]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def timed(fn):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
SHAPES = [(256, 3, 224, 64, 7, 2, 3), (256, 64, 56, 64, 3, 1, 1), (256, 64, 56, 128, 3, 2, 1), (256, 128, 28, 128, 3, 1, 1),
          (256, 64, 56, 128, 1, 2, 0), (256, 256, 14, 256, 3, 1, 1), (256, 512, 7, 512, 3, 1, 1),
          (128, 3, 32, 64, 3, 1, 1), (128, 64, 32, 128, 3, 1, 1), (128, 512, 4, 512, 3, 1, 1)]
for dt in (torch.float32, torch.bfloat16):
    for B, C, H, Co, k, s, p in SHAPES:
        x = torch.randn(B, C, H, H, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
        wf = w.to(dt)
        alpha = torch.rand(Co, device="cuda") * 0.01
        bias = torch.randn(Co, device="cuda")
        ours = lambda: N.conv2d_w8(x, w, s, p, bias=bias, alpha=alpha)
        lib = lambda: N.epilogue_channels_f32(F.conv2d(x, wf, stride=s, padding=p).float().contiguous(), bias, alpha, 0.0)
        to, tl = [], []
        for _ in range(rounds):
            to.append(timed(ours))
            tl.append(timed(lib))
        o, l = sorted(to)[rounds // 2], sorted(tl)[rounds // 2]
        Ho = (H + 2 * p - k) // s + 1
        fl = 2.0 * B * Co * Ho * Ho * C * k * k
        print(f"{str(dt).split('.')[-1]:8s} B={B} {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s}: miq_conv2d_w8 {o:8.1f} us ({fl / o / 1e6:6.1f} TFLOP/s) | "
              f"library conv + epilogue {l:8.1f} us -> {l / o:5.2f}x", flush=True)
