#!/usr/bin/env python3
"""The NCHW entry point on 3x3 / stride 1 layers: the patch kernel staging the NCHW activation itself against the
re-layout launch + patch kernel (developer switch MIQ_CONV_DIRECT=0), ResNet-18's 64->64 56x56 at batch 256 (and
128->128 28x28, which the library keeps on the re-layout path: both columns are that path), whole calls, median of 5
interleaved rounds, microseconds."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for C, H, Co in [(64, 56, 64), (128, 28, 128)]:
    x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Co, device="cuda") * 1e-4
    for od in (torch.int32, torch.float32, torch.bfloat16):
        outs = [torch.empty(256, Co, H, H, dtype=od, device="cuda") for _ in range(3)]
        res = {}
        for _ in range(5):
            for v in ("1", "0"):
                os.environ["MIQ_CONV_DIRECT"] = v
                for i in range(3):
                    N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(20):
                    N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{C}->{Co} {H}x{H} {str(od).split('.')[-1]:8s}: staged from NCHW {statistics.median(res['1']):6.1f} | "
              f"re-layout launch + kernel {statistics.median(res['0']):6.1f}", flush=True)
