#!/usr/bin/env python3
"""Mid-size GEMM timing for tile experiments, one process, the forced tiles (MIQ_GEMM_TILE letters) interleaved:
    python tools/mid_bench.py [letters, default "-eopq"]      ("-" = the library's own choice)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

letters = sys.argv[1] if len(sys.argv) > 1 else "-eopq"
shapes = [(256, 14336, 4096), (512, 14336, 4096), (1024, 14336, 4096), (512, 4096, 4096), (1024, 4096, 4096),
          (2048, 4096, 4096), (256, 28672, 8192), (512, 28672, 8192), (1024, 8192, 8192)]
for M, Nn, K in shapes:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
    al = torch.rand(Nn, device="cuda") * 0.01
    out = torch.empty(M, Nn, device="cuda")
    best = {}
    for rnd in range(3):
        for t in letters:
            if t == "-":
                os.environ.pop("MIQ_GEMM_TILE", None)
            else:
                os.environ["MIQ_GEMM_TILE"] = t
            for _ in range(20):
                N.gemm_i8(a, w, alpha=al, out=out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                N.gemm_i8(a, w, alpha=al, out=out)
            e.record()
            torch.cuda.synchronize()
            best[t] = min(best.get(t, 1e9), s.elapsed_time(e) / 20 * 1e3)
    print(f"M={M} N={Nn} K={K}: " + "  ".join(f"{t} {v:.1f}" for t, v in best.items()) + " us", flush=True)
