#!/usr/bin/env python3
"""Developer micro-benchmarks for the individual kernels (not the driver's bench.py).

    python tools/kbench.py [quant] [gemm] [conv] [small]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402


def timeit(fn, iters=20, warmup=5, ramp_s=0.15):
    """Median and minimum of `iters` single launches timed with HIP events, after `ramp_s` seconds of the
    same launches: the chip needs tens of milliseconds of load to leave its idle clocks."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def rand_i8(shape):
    return torch.randint(-128, 128, shape, dtype=torch.int32, device="cuda").to(torch.int8)


def bench_quant():
    for rows, cols, dt in [(4096, 4096, torch.float32), (14336, 4096, torch.float32), (4096, 14336, torch.float32),
                           (28672, 8192, torch.float32), (8192, 28672, torch.float32),
                           (14336, 4096, torch.bfloat16), (28672, 8192, torch.float16)]:
        nbuf = max(2, int(2e9 // (rows * cols * 4)))
        ws = [torch.randn(rows, cols, device="cuda", dtype=torch.float32).to(dt) * 0.02 for _ in range(nbuf)]
        i = [0]

        def fn():
            N.quantize_rows_sym_i8(ws[i[0] % nbuf])
            i[0] += 1
        med, mn = timeit(fn, iters=3 * nbuf)
        es = ws[0].element_size()
        byts = rows * cols * (es + 1) + rows * 12
        print(f"quant {rows}x{cols} {str(dt)[6:]}: med {med*1e3:.1f} us  {byts/med/1e6:.0f} GB/s (best {byts/mn/1e6:.0f})", flush=True)
        del ws


def bench_gemm(shapes=None):
    shapes = shapes or [(4096, 4096, 4096), (8192, 8192, 8192), (256, 14336, 4096), (256, 4096, 14336),
                        (8192, 28672, 8192), (2048, 8192, 8192)]
    for M, Nn, K in shapes:
        a, w = rand_i8((M, K)), rand_i8((Nn, K))
        alpha = torch.rand(Nn, device="cuda") * 0.01
        bias = torch.randn(Nn, device="cuda")
        for od in (torch.float32, torch.bfloat16, torch.int32):
            out = torch.empty(M, Nn, dtype=od, device="cuda")
            fn = lambda: N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=0.0, out_dtype=od, out=out)  # noqa: E731
            med, mn = timeit(fn)
            ops = 2.0 * M * Nn * K
            print(f"gemm M={M} N={Nn} K={K} out={str(od)[6:]}: med {med*1e3:.1f} us  {ops/med/1e9:.0f} TOPS (best {ops/mn/1e9:.0f})", flush=True)


def bench_small():
    for M in (1, 8, 16, 32, 64, 128, 256):
        for Nn, K in ((14336, 4096), (4096, 14336)):
            a, w = rand_i8((M, K)), rand_i8((Nn, K))
            alpha = torch.rand(Nn, device="cuda") * 0.01
            out = torch.empty(M, Nn, dtype=torch.float32, device="cuda")
            fn = lambda: N.gemm_i8(a, w, alpha=alpha, out=out)  # noqa: E731
            med, mn = timeit(fn)
            print(f"gemm M={M} N={Nn} K={K}: med {med*1e3:.1f} us  {2.0*M*Nn*K/med/1e9:.1f} TOPS  weight {Nn*K/med/1e6:.0f} GB/s", flush=True)


def bench_conv():
    cfgs = [(256, 3, 224, 64, 7, 2, 3), (256, 64, 56, 64, 3, 1, 1), (256, 64, 56, 128, 3, 2, 1), (256, 128, 28, 128, 3, 1, 1),
            (256, 64, 56, 128, 1, 2, 0), (256, 128, 28, 256, 3, 2, 1), (256, 256, 14, 256, 3, 1, 1),
            (256, 256, 14, 512, 3, 2, 1), (256, 512, 7, 512, 3, 1, 1)]
    for B, C, H, Co, k, s, p in cfgs:
        x, w = rand_i8((B, C, H, H)), rand_i8((Co, C, k, k))
        alpha = torch.rand(Co, device="cuda") * 0.01
        Ho = (H + 2 * p - k) // s + 1
        for od in (torch.int32, torch.bfloat16):
            outs = [torch.empty(B, Co, Ho, Ho, dtype=od, device="cuda") for _ in range(3)]   # rotate, as a network does
            cnt = [0]

            def fn():
                cnt[0] += 1
                N.conv2d_i8(x, w, s, p, alpha=alpha, out_dtype=od, out=outs[cnt[0] % 3])
            med, mn = timeit(fn, iters=10, warmup=3)
            ops = 2.0 * B * Co * Ho * Ho * C * k * k
            print(f"conv B={B} {C}->{Co} {H}x{H} k{k}s{s}p{p} out={str(od)[6:]}: med {med*1e3:.1f} us  {ops/med/1e9:.0f} TOPS", flush=True)


def bench_act():
    from quantization_amd.activation import linear_dynamic_w8a8, quantize_tensor_i8
    for shape, dt in (((8192, 8192), torch.float32), ((8192, 8192), torch.bfloat16), ((256, 4096), torch.float32)):
        xs = [torch.randn(shape, device="cuda").to(dt) for _ in range(5)]   # rotate past the cache for the big case
        cnt = [0]

        def mm():
            cnt[0] += 1
            N.tensor_minmax(xs[cnt[0] % 5])

        def qt():
            cnt[0] += 1
            quantize_tensor_i8(xs[cnt[0] % 5])
        med, _ = timeit(mm)
        nb = xs[0].numel() * xs[0].element_size()
        print(f"minmax {shape} {str(dt)[6:]}: med {med*1e3:.1f} us  {nb/med/1e6:.0f} GB/s", flush=True)
        med, _ = timeit(qt)
        print(f"quantize_tensor {shape} {str(dt)[6:]} (min/max + params + quantize): med {med*1e3:.1f} us  "
              f"{(2*nb + xs[0].numel())/med/1e6:.0f} GB/s (2 reads + 1 int8 write)", flush=True)
    for M in (16, 256, 2048):
        x = torch.randn(M, 4096, device="cuda")
        wq, ws, _ = N.quantize_rows_sym_i8(torch.randn(14336, 4096, device="cuda") * 0.02)
        rs = wq.sum(dim=1, dtype=torch.int32).float()
        med, _ = timeit(lambda: linear_dynamic_w8a8(x, wq, ws, None, rs))
        print(f"linear_dynamic_w8a8 M={M} [14336,4096]: med {med*1e3:.1f} us", flush=True)


def bench_pool():
    for mode, dt, shape, k, st, p in (("max", torch.int8, (256, 64, 112, 112), 3, 2, 1), ("max", torch.float32, (256, 64, 112, 112), 3, 2, 1),
                                      ("avg", torch.float32, (256, 128, 56, 56), 2, 2, 0), ("avg", torch.int8, (256, 512, 7, 7), 7, 1, 0)):
        x = torch.randint(-128, 128, shape, dtype=torch.int32, device="cuda").to(dt)
        y = N.pool2d_to_i8(x, k, st, p, mode)
        med, mn = timeit(lambda: N.pool2d_to_i8(x, k, st, p, mode))
        nbytes = x.numel() * x.element_size() + y.numel()
        print(f"pool {mode} {str(dt)[6:]} {shape} k{k}s{st}p{p}: med {med*1e3:.1f} us  {nbytes/med/1e6:.0f} GB/s (in + out bytes)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["quant", "gemm", "conv", "small", "pool", "act"]
    t0 = time.time()
    if "quant" in which:
        bench_quant()
    if "gemm" in which:
        bench_gemm()
    if "small" in which:
        bench_small()
    if "conv" in which:
        bench_conv()
    if "pool" in which:
        bench_pool()
    if "act" in which:
        bench_act()
    print(f"done in {time.time()-t0:.1f}s")
