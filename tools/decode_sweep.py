#!/usr/bin/env python3
"""Decode-size int8 GEMM (M <= 16) with the weight streamed from HBM: the launches rotate over enough copies of
the weight to exceed the 256 MiB Infinity Cache, as the layers of a model do.  Sweeps the developer variants of
the weight-streaming kernel (MIQ_SKINNY = channel groups per workgroup, K-tiles in flight per wave, n = non-temporal
weight loads) against the library's own choice.  Needs the developer library.

    python tools/decode_sweep.py [M ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

Ms = [int(v) for v in sys.argv[1:]] or [1, 16]
VARIANTS = ["auto", "skinny", "14c", "14n", "18c", "18n", "24c", "24n", "28n", "12n", "22n", "42n", "44n"]


def run(fn, launches=48):
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / launches * 1e3)
    return sorted(best)[len(best) // 2]


for Nn, K in ((14336, 4096), (4096, 14336), (28672, 4096)):
    ncopy = max(2, (600 << 20) // (Nn * K))
    ws = [torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8) for _ in range(ncopy)]
    alpha = torch.rand(Nn, device="cuda") * 0.01
    for M in Ms:
        a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
        out = torch.empty(M, Nn, device="cuda")
        ref = None
        line = []
        for v in VARIANTS:
            os.environ.pop("MIQ_SKINNY", None)
            os.environ.pop("MIQ_GEMM_SKINNY", None)
            if v == "skinny":
                os.environ["MIQ_GEMM_SKINNY"] = "1"
            elif v != "auto":
                os.environ["MIQ_GEMM_SKINNY"] = "1"
                os.environ["MIQ_SKINNY"] = v
            c = N.gemm_i8(a, ws[0], alpha=alpha).clone()
            if ref is None:
                ref = c
            ok = torch.equal(c, ref)
            it = [0]

            def fn():
                it[0] += 1
                N.gemm_i8(a, ws[it[0] % ncopy], alpha=alpha, out=out)
            us = run(fn)
            line.append(f"{v}{'' if ok else '(WRONG)'} {us:.1f}us/{Nn * K / us / 1e6:.2f}TB/s")
        print(f"[{Nn},{K}] M={M} ({ncopy} weight copies): " + "  ".join(line), flush=True)
    del ws
