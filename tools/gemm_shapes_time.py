#!/usr/bin/env python3
"""HIP-graph timing of the int8 GEMM on the shapes the bench reports (fp32 and bf16 out), rotating operands.

    [MIQ_LIB=path/to/libmiq.so] python tools/gemm_shapes_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

if os.environ.get("MIQ_LIB"):   # another build of the library, for A/B runs on one box
    N.LIB_PATH = os.path.abspath(os.environ["MIQ_LIB"])

dev = torch.device("cuda", 0)
SHAPES = [(8192, 8192, 8192), (4096, 4096, 4096), (8192, 28672, 8192), (2048, 8192, 8192), (8192, 8192, 28672), (1024, 8192, 8192)]
for M, Nn, K in SHAPES:
    for od in (torch.float32, torch.bfloat16):
        nrot = max(2, min(6, int(600e6 // (Nn * K))))
        xs = [torch.randint(-128, 128, (M, K), dtype=torch.int32, device=dev).to(torch.int8) for _ in range(2)]
        ws = [torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device=dev).to(torch.int8) for _ in range(nrot)]
        alpha = torch.rand(Nn, device=dev) * 1e-3
        bias = torch.randn(Nn, device=dev)
        outs = [torch.empty(M, Nn, dtype=od, device=dev) for _ in range(2)]
        calls = 12
        for i in range(30):
            N.gemm_i8(xs[i % 2], ws[i % nrot], bias=bias, alpha=alpha, out_dtype=od, out=outs[i % 2])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(calls):
                N.gemm_i8(xs[i % 2], ws[i % nrot], bias=bias, alpha=alpha, out_dtype=od, out=outs[i % 2])
        for _ in range(5):
            g.replay()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / (4 * calls) * 1e3)
        ts.sort()
        us = ts[len(ts) // 2]
        print(f"{M} x {Nn} x {K} {str(od)[6:]:8s}: {us:8.1f} us  {2 * M * Nn * K / us / 1e6:7.0f} TOPS  frac {2 * M * Nn * K / us / 1e6 / 5000:.4f}", flush=True)
        del xs, ws, outs
