#!/usr/bin/env python3
"""Run the int8 GEMM alone (for rocprofv3 counter passes and data-dependence checks).

    python tools/gemm_probe.py [size] [iters] [data: rand|zero|small] [out: f32|bf16|i32]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data = sys.argv[3] if len(sys.argv) > 3 else "rand"
outk = sys.argv[4] if len(sys.argv) > 4 else "f32"
od = {"f32": torch.float32, "bf16": torch.bfloat16, "i32": torch.int32}[outk]
M = Nn = K = size
if data == "zero":
    a = torch.zeros(M, K, dtype=torch.int8, device="cuda")
    w = torch.zeros(Nn, K, dtype=torch.int8, device="cuda")
elif data == "small":
    a = torch.randint(-3, 4, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-3, 4, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
else:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Nn, device="cuda") * 0.01
out = torch.empty(M, Nn, dtype=od, device="cuda")
for _ in range(3):
    N.gemm_i8(a, w, alpha=alpha, out_dtype=od, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    N.gemm_i8(a, w, alpha=alpha, out_dtype=od, out=out)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"gemm {size}^3 data={data} out={outk}: {ms*1e3:.1f} us  {2.0*M*Nn*K/ms/1e9:.0f} TOPS", flush=True)
