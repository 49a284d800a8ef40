#!/usr/bin/env python3
"""Weight-only Conv2d on floating-point activations: the LDS-resident-patch kernel for 3 x 3 / stride 1 layers against the
im2col-order kernel (developer switch MIQ_CONV_W8_PATCH=0 of libmiq_dev.so), ResNet-18's layers at batch 256, whole calls
(re-layout launches included), fp32 out, one process, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B, dev = 256, torch.device("cuda", 0)
for dt in (torch.float32, torch.bfloat16):
    tot = [0.0, 0.0]
    for C, H, Co, times in ((64, 56, 64, 4), (128, 28, 128, 3), (256, 14, 256, 3), (512, 7, 512, 3)):
        x = torch.randn(B, C, H, H, device=dev).to(dt)
        w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device=dev).to(torch.int8)
        alpha = torch.rand(Co, device=dev) * 1e-3
        outs = [torch.empty(B, Co, H, H, device=dev) for _ in range(3)]
        it = [0]

        def call():
            N.conv2d_w8(x, w, 1, 1, alpha=alpha, out=outs[it[0] % 3])
            it[0] += 1
        r = []
        for v in ("0", None, "0", None):
            if v is None:
                os.environ.pop("MIQ_CONV_W8_PATCH", None)
            else:
                os.environ["MIQ_CONV_W8_PATCH"] = v
            r.append(bench.median_ms(call, 10, 5) * 1e3)
        a, b = min(r[0], r[2]), min(r[1], r[3])
        tot[0] += a * times
        tot[1] += b * times
        fl = 2.0 * B * Co * H * H * C * 9
        print(f"{str(dt).split('.')[-1]:8s} {C:3d}->{Co:3d} {H:2d}x{H:<2d} x{times}: im2col order {a:7.1f} us | patch {b:7.1f} us ({fl / b / 1e6:5.0f} TFLOP/s)", flush=True)
    print(f"{str(dt).split('.')[-1]:8s} these 13 layers: {tot[0]:.0f} us | {tot[1]:.0f} us", flush=True)
