#!/usr/bin/env python3
"""Tile sweep of miq_gemm_w8 through its developer switches (MIQ_W8F_BM / MIQ_W8F_BN, read per call), interleaved
rounds in one process.  Prints the time of every tile shape per case and marks what the default plan gives.

    python tools/w8f_sweep.py [f32|bf16] [rounds] [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20


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


def setenv(bm, bn):
    for k, v in (("MIQ_W8F_BM", bm), ("MIQ_W8F_BN", bn)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


CASES = [(m, 14336, 4096) for m in (1, 8, 16, 32, 48, 64, 96, 128, 192, 256, 512)] + \
        [(m, 4096, 14336) for m in (1, 16, 32, 64, 128, 256)]
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(50):
    spin @ spin
for M, Nn, K in CASES:
    x = torch.randn(M, K, device="cuda").to(dt)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Nn, device="cuda") * 0.01
    out = torch.empty(M, Nn, device="cuda")
    fn = lambda: N.gemm_w8(x, w, alpha=alpha, out=out)
    variants = [(None, None)]
    for bm in (16, 32, 64, 128):
        if bm < 128 and M > 4 * bm:
            continue
        if bm > 16 and M <= bm // 2:
            continue
        variants += [(bm, bn) for bn in ((64, 128, 256) if bm < 128 else (None,))]
    res = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            setenv(*v)
            res[v].append(timed(fn))
    setenv(None, None)
    line = []
    for v in variants:
        t = sorted(res[v])[len(res[v]) // 2]
        line.append(("default" if v == (None, None) else f"{v[0]}x{v[1] or 128}") + f" {t:6.1f}")
    print(f"M={M:4d} [{Nn},{K}] {str(dt).split('.')[-1]}: " + " | ".join(line), flush=True)
