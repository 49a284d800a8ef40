#!/usr/bin/env python3
"""Tile-count quantisation of the 256 x 256 int8 kernel: shapes whose tile count is a little over a multiple of the
256 CUs, default plan against the smaller-tile kernels (developer switches), interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def setenv(plan256, tile, rowsplit=None):
    for k, v in (("MIQ_GEMM_PLAN256", plan256), ("MIQ_GEMM_TILE", tile), ("MIQ_GEMM_ROWSPLIT", rowsplit)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for M, Nn, K in [(2048, 8192, 8192), (2100, 8192, 8192), (2304, 8192, 8192), (2560, 8192, 8192), (3072, 8192, 8192), (4096, 8192, 8192),
                 (4352, 8192, 8192), (1280, 14336, 4096), (1536, 14336, 4096), (2048, 14336, 4096)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Nn, device="cuda") * 0.01
    out = torch.empty(M, Nn, device="cuda")
    fn = lambda: N.gemm_i8(a, w, alpha=alpha, out=out)
    variants = {"default": (None, None), "uncut": (None, None, "0"), "128x128": ("0", "a"), "128x128 8w": ("0", "d"), "256x128": ("0", "e"), "128x256 ring": ("0", "p")}
    res = {k: [] for k in variants}
    for _ in range(3):
        for k, v in variants.items():
            setenv(*v)
            res[k].append(timed(fn))
    setenv(None, None, None)
    t256 = -(-M // 256) * -(-Nn // 256)
    print(f"{M}x{Nn}x{K} ({t256} tiles of 256x256): " + " | ".join(f"{k} {sorted(t)[1]:7.1f}" for k, t in res.items()), flush=True)
