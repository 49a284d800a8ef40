#!/usr/bin/env python3
"""The 8192^3 int8 GEMM on operands of different statistics, one after the other in one process, for a
rocprofv3 counter pass: GRBM_GUI_ACTIVE / 8 / duration is the clock the chip held, SQ_VALU_MFMA_BUSY_CYCLES
the cycles its matrix pipes were busy.  Kernel durations come from the same run's kernel trace, in launch
order: zero, small (|v| <= 3), gauss (sigma 32, what the quantizer makes of normal weights) x uniform, uniform.

    python tools/gemm_data_probe.py [size] [launches] [case ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
g = torch.Generator(device="cuda").manual_seed(0)
uni = lambda: torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
cases = {
    "zero": (torch.zeros(size, size, dtype=torch.int8, device="cuda"), torch.zeros(size, size, dtype=torch.int8, device="cuda")),
    "small": (torch.randint(-3, 4, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8),
              torch.randint(-3, 4, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)),
    "gauss": (uni(), (torch.randn(size, size, generator=g, device="cuda") * 32).round().clamp(-128, 127).to(torch.int8)),
    "uniform": (uni(), uni()),
}
alpha = torch.rand(size, generator=g, device="cuda") * 1e-3
out = torch.empty(size, size, device="cuda")
only = sys.argv[3:]
for name, (a, w) in cases.items():
    if only and name not in only:
        continue
    t_end = time.perf_counter() + 0.4
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
    print(f"{name}: {us:.1f} us  {2.0*size**3/us/1e6:.0f} TOPS", flush=True)
