#!/usr/bin/env python3
"""A/B of the weight-only 256 x 256 kernel's two schedules (developer switch MIQ_W8F_SCHED of libmiq_dev.so: 0 = four
phases of 16 MFMAs, default = two segments of 32 with the k-halves of an accumulator back to back), interleaved rounds in
one process, results compared bit for bit:  python tools/w8f_sched_ab.py [rounds] [iters]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20


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


def sched(v):
    if v is None:
        os.environ.pop("MIQ_W8F_SCHED", None)
    else:
        os.environ["MIQ_W8F_SCHED"] = v


for dt in (torch.bfloat16, torch.float32):
    for M, Nn, K in [(8192, 8192, 8192), (4096, 4096, 4096), (2048, 14336, 4096)]:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(Nn, device="cuda") * 0.01
        out = torch.empty(M, Nn, device="cuda")
        fn = lambda: N.gemm_w8(x, w, alpha=alpha, out=out)
        sched("0")
        ref = N.gemm_w8(x, w, alpha=alpha).clone()
        sched(None)
        same = torch.equal(ref, N.gemm_w8(x, w, alpha=alpha))
        t = {"0": [], None: []}
        for _ in range(rounds):
            for v in ("0", None):
                sched(v)
                t[v].append(timed(fn))
        a, b = statistics.median(t["0"]), statistics.median(t[None])
        fl = 2.0 * M * Nn * K
        print(f"{str(dt).split('.')[-1]:8s} {M}x{Nn}x{K}: four phases {a:7.1f} us ({fl / a / 1e6:5.0f} TF) | two segments {b:7.1f} us "
              f"({fl / b / 1e6:5.0f} TF)  {'bit-equal' if same else 'DIFFERENT'}", flush=True)
