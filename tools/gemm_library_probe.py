#!/usr/bin/env python3
"""Calibration: the vendor library's int8 GEMM (torch._int_mm -> hipBLASLt, int32 out, no epilogue) beside
miq_gemm_i8 (int32 out and the fused fp32 epilogue) on the same random int8 operands, interleaved in one
process.  Says whether the rate the chip sustains under this library's kernel is a property of the kernel or of
the part.  Run under `rocprofv3 --kernel-trace --stats` to see which library kernel (tile, MFMA shape) ran.

    python tools/gemm_library_probe.py [size ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

sizes = [int(s) for s in sys.argv[1:]] or [8192, 4096]
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, launches=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches * 1e3


for size in sizes:
    a = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    wt = w.t()  # [K, N] view with K contiguous per output channel: the library's "NT" form
    alpha = torch.rand(size, generator=g, device="cuda") * 1e-3
    out_f = torch.empty(size, size, device="cuda")
    out_i = torch.empty(size, size, dtype=torch.int32, device="cuda")
    cases = {
        "miq fp32 out": lambda: N.gemm_i8(a, w, alpha=alpha, out=out_f),
        "miq int32 out": lambda: N.gemm_i8(a, w, out_dtype=torch.int32, out=out_i),
        "library int32 out (w as [N,K].t())": lambda: torch._int_mm(a, wt, out=out_i),
        "library int32 out (w as [K,N])": lambda: torch._int_mm(a, w, out=out_i),
    }
    ref = torch._int_mm(a, wt)
    N.gemm_i8(a, w, out_dtype=torch.int32, out=out_i)
    print(f"size {size}: bit-equal to the library: {bool(torch.equal(ref, out_i))}", flush=True)
    del ref
    best = {k: 1e9 for k in cases}
    for rnd in range(3):
        for name, fn in cases.items():
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                fn()
                torch.cuda.synchronize()
            best[name] = min(best[name], timed(fn))
    for name, us in best.items():
        print(f"  {name}: {us:.1f} us  {2.0*size**3/us/1e6:.0f} TOPS", flush=True)
