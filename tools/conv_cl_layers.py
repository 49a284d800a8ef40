#!/usr/bin/env python3
"""Per-layer timings of ResNet-18's convolutions at batch 256 on the channels-last int8 path (miq_conv2d_i8_nhwc, int8
out with the next layer's border), beside the NCHW / int32 contract path (miq_conv2d_i8): whole calls, HIP events,
median of 30, outputs rotating over three buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B, dev = 256, torch.device("cuda", 0)
tot = [0.0, 0.0]
for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
    Cc = 4 if C == 3 else C
    xn = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
    wn = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    x = torch.zeros(B, H + 2 * p, H + 2 * p, Cc, dtype=torch.int8, device=dev)
    x[:, p:H + p, p:H + p, :C] = xn.permute(0, 2, 3, 1)
    w = torch.zeros(Co, Cc, k, k, dtype=torch.int8, device=dev)
    w[:, :C] = wn
    wp = N.conv2d_i8_pack_weight(w)
    alpha = torch.rand(Co, device=dev) * 1e-4
    Ho = (H + 2 * p - k) // s + 1
    outs = [torch.zeros(B, Ho + 2, Ho + 2, Co, dtype=torch.int8, device=dev) for _ in range(3)]
    outn = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
    it = [0]

    def cl():
        N.conv2d_i8_nhwc(x, wp, w.shape, s, p, alpha=alpha, out_dtype=torch.int8, out_pad=1, out=outs[it[0] % 3])
        it[0] += 1

    def nchw():
        N.conv2d_i8(xn, wn, s, p, out_dtype=torch.int32, out=outn[it[0] % 3])
        it[0] += 1
    a, b = bench.median_ms(cl, 30, 30) * 1e3, bench.median_ms(nchw, 30, 30) * 1e3
    ops = 2.0 * B * Co * Ho * Ho * C * k * k
    tot[0] += a * times
    tot[1] += b * times
    print(f"conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s} x{times}: channels-last int8 {a:6.1f} us ({ops / a / 1e6:5.0f} TOPS) | "
          f"NCHW int32 {b:6.1f} us", flush=True)
print(f"stack: channels-last int8 {tot[0]:.0f} us | NCHW int32 {tot[1]:.0f} us")
