#!/usr/bin/env python3
"""Tiles of 64 pixels per workgroup of the int8 convolution's re-layout launch (developer switch MIQ_CONV_PREP_NT = 1, 2, 4),
whole calls of the ResNet-18 layers that have such a launch, batch 256, interleaved in one process:
    python tools/conv_prep_nt_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B = 256
dev = torch.device("cuda", 0)
for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
    if C < 64 or (C == 64 and k == 3 and s == 1):
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
    line = []
    for nt in ("1", "2", "4", "1", "2", "4"):
        os.environ["MIQ_CONV_PREP_NT"] = nt
        line.append(f"{nt}: {bench.graphed_ms(step, 6, 10, 20) * 1e3:6.1f}")
    print(f"conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s}: " + "  ".join(line) + " us", flush=True)
