#!/usr/bin/env python3
"""Large-M tile A/B of miq_gemm_w8 (developer switches, interleaved rounds in one process) beside the library
bf16 GEMM:  python tools/w8f_big.py [rounds] [iters]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def timed(fn):
    for _ in range(2):
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


spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for dt in (torch.bfloat16, torch.float32):
    for M, Nn, K in [(512, 14336, 4096), (2048, 14336, 4096), (2048, 4096, 14336), (8192, 8192, 8192)]:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
        wf = w.to(dt)
        alpha = torch.rand(Nn, device="cuda") * 0.01
        out = torch.empty(M, Nn, device="cuda")
        fn = lambda: N.gemm_w8(x, w, alpha=alpha, out=out)
        variants = [(None, None), (128, 128), (64, 256), (256, 256)] + ([(128, 256)] if dt == torch.bfloat16 else [])
        res = {v: [] for v in variants}
        lib = []
        for _ in range(rounds):
            for v in variants:
                setenv(*v)
                res[v].append(timed(fn))
            setenv(None, None)
            lib.append(timed(lambda: F.linear(x, wf)))
        flops = 2.0 * M * Nn * K
        line = [("default" if v == (None, None) else f"{v[0]}x{v[1]}") + f" {sorted(res[v])[rounds // 2]:7.1f} us ({flops / sorted(res[v])[rounds // 2] / 1e6:5.0f} TF)"
                for v in variants]
        print(f"{str(dt).split('.')[-1]:8s} M={M:5d} [{Nn},{K}]: " + " | ".join(line) + f" | library GEMM alone {sorted(lib)[rounds // 2]:7.1f} us", flush=True)
