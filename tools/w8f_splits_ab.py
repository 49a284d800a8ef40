#!/usr/bin/env python3
"""A/B of forced split-K counts (developer switch MIQ_W8F_SPLITS) for decode sizes of miq_gemm_w8 on the Llama-3-8B
gate/up weight [14336,4096]: whole calls, HIP events, median of 5 interleaved rounds.

    python tools/w8f_splits_ab.py [M, default 16]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Nn, K = 14336, 4096
w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Nn, device="cuda") * 0.01
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(M, K, device="cuda").to(dt)
    out = torch.empty(M, Nn, device="cuda")
    res, ref = {}, None
    for r in range(5):
        for v in ("1", "2", "3", "4"):
            os.environ["MIQ_W8F_SPLITS"] = v
            for _ in range(10):
                N.gemm_w8(x, w, alpha=alpha, out=out)
            if r == 0:
                if ref is None:
                    ref = out.clone()
                else:
                    print(f"splits {v}: max |diff| vs 1 = {float((out - ref).abs().max()):.3e}", flush=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                N.gemm_w8(x, w, alpha=alpha, out=out)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"M={M} {str(dt)[6:]}: " + " | ".join(f"splits {v} {statistics.median(t):6.1f} us" for v, t in res.items()), flush=True)
