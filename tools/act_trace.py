#!/usr/bin/env python3
"""The dynamic activation quantizer's launches for a rocprofv3 --kernel-trace run (tools/trace_medians.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402
from quantization_amd.activation import linear_dynamic_w8a8, quantize_tensor_i8  # noqa: E402

wq, ws, _ = N.quantize_rows_sym_i8(torch.randn(14336, 4096, device="cuda") * 0.02)
rs = wq.sum(dim=1, dtype=torch.int32).float()
for M in (16, 256):
    x = torch.randn(M, 4096, device="cuda")
    for _ in range(30):
        linear_dynamic_w8a8(x, wq, ws, None, rs)
    torch.cuda.synchronize()
x = torch.randn(256, 4096, device="cuda")
for _ in range(30):
    quantize_tensor_i8(x)
torch.cuda.synchronize()
