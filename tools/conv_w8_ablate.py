#!/usr/bin/env python3
"""Timing ablations of the weight-only patch kernel (developer library, MIQ_W8_PATCH_ABLATE: 1 = no patch fetch traffic,
2 = no output stores, 3 = both; results are wrong on purpose).  Kernel time from a HIP-graph of whole calls minus nothing:
the re-layout launch is inside every number.    python tools/conv_w8_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

B = 256
for C, H, Co in ((64, 56, 64), (128, 28, 128), (256, 14, 256)):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, C, H, H, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(Co, device="cuda") * 1e-3
        pk = N.conv2d_w8_prepack(w, x.dtype)
        outs = [torch.empty(B, Co, H, H, device="cuda") for _ in range(3)]
        it = [0]

        def call():
            N.conv2d_w8(x, w, 1, 1, alpha=alpha, out=outs[it[0] % 3], packed=pk)
            it[0] += 1
        line = []
        for mode in ("0", "1", "2", "3"):
            os.environ["MIQ_W8_PATCH_ABLATE"] = mode
            line.append(f"{bench.median_ms(call, 10, 5) * 1e3:7.1f}")
        print(f"{C}->{Co} {H}x{H} {str(dt)[6:]:8s} whole call us: full {line[0]} | no fetch traffic {line[1]} | no stores {line[2]} | neither {line[3]}", flush=True)
