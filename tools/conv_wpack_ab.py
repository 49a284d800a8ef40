#!/usr/bin/env python3
"""What the per-call filter re-layout costs the NCHW int8 convolution: ResNet-18's layers at batch 256, int32 out, whole
calls, with and without the re-layout of the filters (developer switch MIQ_CONV_SKIP_WPACK of libmiq_dev.so: timing-only,
the workspace block the allocator hands back still holds the previous call's packed filters)."""
import os
import sys

import torch

import _devlib  # noqa: F401
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B, dev = 256, torch.device("cuda", 0)
tot = [0.0, 0.0]
for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
    xn = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
    wn = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    Ho = (H + 2 * p - k) // s + 1
    outn = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
    it = [0]

    def nchw():
        N.conv2d_i8(xn, wn, s, p, out_dtype=torch.int32, out=outn[it[0] % 3])
        it[0] += 1
    r = []
    for skip in (False, True, False, True):
        if skip:
            os.environ["MIQ_CONV_SKIP_WPACK"] = "1"
        else:
            os.environ.pop("MIQ_CONV_SKIP_WPACK", None)
        r.append(bench.median_ms(nchw, 30, 30) * 1e3)
    a, b = min(r[0], r[2]), min(r[1], r[3])
    tot[0] += a * times
    tot[1] += b * times
    print(f"conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s} x{times}: with filter re-layout {a:6.1f} us | without {b:6.1f} us", flush=True)
print(f"stack: with {tot[0]:.0f} us | without {tot[1]:.0f} us")
