#!/usr/bin/env python3
"""The reference's CPU path (oracle/oracle_torch.py, pinned to the reference by fixtures) timed on
this machine's host cores: quantizer, weight-only forward, an int32
matmul standing in for linear_w8a8, and a ResNet-18 layer as F.conv2d.

    python tools/cpu_baseline.py [threads]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_torch as T  # noqa: E402


def best_of(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    if len(sys.argv) > 1:
        torch.set_num_threads(int(sys.argv[1]))
    print(f"threads={torch.get_num_threads()}", flush=True)
    g = torch.Generator().manual_seed(0)
    for rows, cols in [(14336, 4096), (28672, 8192)]:
        w = torch.randn(rows, cols, generator=g) * 0.02
        t = best_of(lambda: T.quantize_channel_sym_int8(w), 2)
        print(f"quantize [{rows},{cols}] fp32: {t:.3f} s  {rows*cols*5/t/1e9:.2f} GB/s", flush=True)
    for n, k in [(14336, 4096), (4096, 14336), (8192, 8192)]:
        wq = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int32).to(torch.int8)
        scale, zp, bias = torch.rand(n, generator=g) * 0.01, torch.zeros(n, dtype=torch.int64), torch.randn(n, generator=g)
        x = torch.randint(-128, 128, (256, k), generator=g, dtype=torch.int32).float()
        t = best_of(lambda: T.wo_linear_forward(x, wq, scale, zp, bias))
        print(f"weight-only forward x[256,{k}] . Wq[{n},{k}]^T: {t*1e3:.1f} ms  {2*256*n*k/t/1e12:.3f} TOPS-equivalent", flush=True)
    a = torch.randint(-128, 128, (1024, 1024), generator=g, dtype=torch.int32)
    b = torch.randint(-128, 128, (1024, 1024), generator=g, dtype=torch.int32)
    t = best_of(lambda: torch.matmul(a, b), 2)
    print(f"int32 matmul 1024^3 (sample of the 4096^3 case): {t*1e3:.1f} ms  {2*1024**3/t/1e12:.4f} TOPS", flush=True)
    x = torch.randint(-128, 128, (256, 64, 56, 56), generator=g, dtype=torch.int32).float()
    w = torch.randint(-128, 128, (64, 64, 3, 3), generator=g, dtype=torch.int32).float()
    t = best_of(lambda: F.conv2d(x, w, stride=1, padding=1))
    print(f"F.conv2d fp32 256x64x56x56 * 64x64x3x3: {t*1e3:.1f} ms  {2*256*64*56*56*64*9/t/1e12:.3f} TOPS-equivalent", flush=True)


if __name__ == "__main__":
    main()
