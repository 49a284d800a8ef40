#!/usr/bin/env python3
"""Does the row pitch of the operands matter?  Times the int8 GEMM at M = N = 8192 for several K around 8192
(the pitch of both operands is K bytes), interleaved rounds in one process, and prints TOPS per K.

    python tools/gemm_kstride.py [rounds] [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M = Nn = 8192
Ks = [8192, 8192 + 128, 8192 + 256, 8192 + 512, 8192 - 128, 8192 + 2048]
cases = []
for K in Ks:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
    cases.append((K, a, w))
alpha = torch.rand(Nn, device="cuda") * 0.01
out = torch.empty(M, Nn, dtype=torch.float32, device="cuda")
best = {K: [] for K in Ks}
for _ in range(40):
    N.gemm_i8(cases[0][1], cases[0][2], alpha=alpha, out=out)
for r in range(rounds):
    for K, a, w in cases:
        for _ in range(3):
            N.gemm_i8(a, w, alpha=alpha, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            N.gemm_i8(a, w, alpha=alpha, out=out)
        e.record()
        torch.cuda.synchronize()
        best[K].append(s.elapsed_time(e) / iters)
for K in Ks:
    t = sorted(best[K])
    med = t[len(t) // 2]
    print(f"K={K:6d}: median {med*1e3:7.1f} us  {2.0*M*Nn*K/med/1e9:6.0f} TOPS   min {t[0]*1e3:7.1f} us", flush=True)
