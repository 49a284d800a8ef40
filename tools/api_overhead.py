#!/usr/bin/env python3
"""Host overhead of the drop-in layer: Quantizer.forward (quantization_amd.quant.Weight_Only) against the bare
C-ABI call, on a decode-size Linear (the kernel takes ~15 us, so anything above that is host time).

    python tools/api_overhead.py [M] [bf16|f32|i8]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402
from quantization_amd.quant.Weight_Only import Quantizer  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kind = sys.argv[2] if len(sys.argv) > 2 else "bf16"
qp = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}
lin = torch.nn.Linear(4096, 14336).cuda()
ql = Quantizer.from_float(module=lin, data_metry=qp)
if kind == "i8":
    x = torch.randint(-128, 128, (M, 4096), dtype=torch.int32, device="cuda").to(torch.int8)
else:
    x = torch.randn(M, 4096, device="cuda").to({"bf16": torch.bfloat16, "f32": torch.float32}[kind])
w, scale, bias = ql.weight, ql.weight_quant.scales.reshape(-1).contiguous(), lin.bias.detach()
out = torch.empty(M, 14336, device="cuda")


def wall(fn, iters=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


raw = (lambda: N.gemm_i8(x, w, bias=bias, alpha=scale, out=out)) if kind == "i8" else \
      (lambda: N.gemm_w8(x, w, bias=bias, alpha=scale, out=out))
print(f"M={M} {kind}: Quantizer.forward {wall(lambda: ql(x)):6.1f} us per call | bare wrapper call {wall(raw):6.1f} us", flush=True)
