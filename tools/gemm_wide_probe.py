#!/usr/bin/env python3
"""Runs the 256 x 384 and the 256 x 256 GEMM kernel on the same problem in one process, for rocprofv3
counter passes (the two kernels have different names, so their counters separate by name).  Needs the
developer library.

    python tools/gemm_wide_probe.py [M N K] [launches]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

M, Nn, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 6144, 8192)
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 40
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (Nn, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Nn, generator=g, device="cuda") * 1e-3
out = torch.empty(M, Nn, device="cuda")
for spec in ("0", str(Nn // 384)):
    os.environ["MIQ_GEMM_WIDE"] = spec
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        N.gemm_i8(a, w, alpha=alpha, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        N.gemm_i8(a, w, alpha=alpha, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / launches * 1e3
    print(f"wide={spec}: {us:.1f} us  {2.0*M*Nn*K/us/1e6:.0f} TOPS", flush=True)
