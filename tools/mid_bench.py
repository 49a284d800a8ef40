#!/usr/bin/env python3
"""Mid-size GEMM timing for tile experiments: python tools/mid_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N
shapes = [(256, 14336, 4096), (256, 4096, 14336), (512, 14336, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (256, 28672, 8192), (1024, 8192, 8192)]
for M, Nn, K in shapes:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
    al = torch.rand(Nn, device="cuda") * 0.01
    out = torch.empty(M, Nn, device="cuda")
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:   # clock ramp
        for _ in range(10): N.gemm_i8(a, w, alpha=al, out=out)
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): N.gemm_i8(a, w, alpha=al, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"M={M} N={Nn} K={K}: {ms*1e3:.1f} us {2.0*M*Nn*K/ms/1e9:.0f} TOPS", flush=True)
