#!/usr/bin/env python3
"""Run miq_conv2d_w8 alone (for rocprofv3 kernel traces): python tools/conv_w8_probe.py B Cin H Cout k s p [f32|bf16] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

B, C, H, Co, k, s, p = (int(v) for v in sys.argv[1:8])
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[8] if len(sys.argv) > 8 else "f32"]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
x = torch.randn(B, C, H, H, device="cuda").to(dt)
w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Co, device="cuda") * 0.01
for _ in range(iters):
    N.conv2d_w8(x, w, s, p, alpha=alpha)
torch.cuda.synchronize()
