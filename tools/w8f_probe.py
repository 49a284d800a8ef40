#!/usr/bin/env python3
"""Run miq_gemm_w8 alone (for rocprofv3 kernel traces):  python tools/w8f_probe.py M N K [f32|bf16|f16] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

M, Nn, K = (int(v) for v in sys.argv[1:4])
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
x = torch.randn(M, K, device="cuda").to(dt)
w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Nn, device="cuda") * 0.01
out = torch.empty(M, Nn, device="cuda")
for _ in range(iters):
    N.gemm_w8(x, w, alpha=alpha, out=out)
torch.cuda.synchronize()
