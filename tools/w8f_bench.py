#!/usr/bin/env python3
"""Weight-only Linear on floating-point activations: miq_gemm_w8 (int8 weight read as stored, bf16 MFMA, fused
epilogue) beside the library route it replaces (float GEMM on a cached float copy of the weight, as the reference's
Quantizer.forward minus its per-call weight cast, then the fused epilogue kernel), interleaved in one process.

    python tools/w8f_bench.py [rounds] [iters]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3] if len(sys.argv) > 3 else ""


def timed(fn):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


CASES = [(m, 14336, 4096) for m in (1, 16, 32, 64, 128, 256)] + [(m, 4096, 14336) for m in (1, 16, 64, 256)] + \
        [(2048, 14336, 4096), (8192, 8192, 8192)]
for dt in (torch.float32, torch.bfloat16, torch.float16):
    if only and only not in str(dt):
        continue
    for M, Nn, K in CASES:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
        wf = w.to(dt)
        alpha = torch.rand(Nn, device="cuda") * 0.01
        bias = torch.randn(Nn, device="cuda")
        out = torch.empty(M, Nn, device="cuda")
        ours = lambda: N.gemm_w8(x, w, bias=bias, alpha=alpha, out=out)
        lib = lambda: N.epilogue_channels_f32(F.linear(x, wf).float(), bias, alpha, 0.0)
        to, tl = [], []
        for _ in range(rounds):
            to.append(timed(ours))
            tl.append(timed(lib))
        to.sort(), tl.sort()
        o, l = to[len(to) // 2], tl[len(tl) // 2]
        flops = 2.0 * M * Nn * K
        print(f"{str(dt).split('.')[-1]:9s} M={M:5d} [{Nn},{K}]: miq_gemm_w8 {o:8.1f} us ({flops/o/1e6:7.1f} TFLOP/s, weight at "
              f"{Nn*K/o/1e3:6.0f} GB/s) | library float GEMM + epilogue {l:8.1f} us  -> {l/o:5.2f}x", flush=True)
