#!/usr/bin/env python3
"""Whole-configuration timings for the BASELINE.json configs that are not the headline line:

  config 3  ResNet-18's twenty convolutions at batch 256 (int8 in, int32 / bf16 out), per layer and in total
  config 4  Llama-3-8B MLP (gate, up [14336,4096]; down [4096,14336]): quantize once, then the three int8
            matmuls per token batch M = 1 .. 256 (fp32 out, fused dequant epilogue), and the same stack on
            bf16 / fp32 activations (the weight-only forward, miq_gemm_w8)
  config 5  Llama-3-70B shapes on ONE GPU ([28672,8192], [8192,28672], [8192,8192]) at M = 256, 2048, 8192
            (the column-sharded run over 2/4/8 GPUs is bench.py --gpus N)

    python tools/config_bench.py [3] [4] [5]

HIP events around each call, median of 15 after a clock ramp; outputs rotate over three buffers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402


def med_us(fn, iters=15, ramp_s=0.1):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


def rand_i8(shape):
    return torch.randint(-128, 128, shape, dtype=torch.int32, device="cuda").to(torch.int8)


def config3():
    # (Cin, H, Cout, k, stride, pad, how many times the layer occurs in ResNet-18)
    layers = [(3, 224, 64, 7, 2, 3, 1), (64, 56, 64, 3, 1, 1, 4), (64, 56, 128, 3, 2, 1, 1), (128, 28, 128, 3, 1, 1, 3),
              (64, 56, 128, 1, 2, 0, 1), (128, 28, 256, 3, 2, 1, 1), (256, 14, 256, 3, 1, 1, 3), (128, 28, 256, 1, 2, 0, 1),
              (256, 14, 512, 3, 2, 1, 1), (512, 7, 512, 3, 1, 1, 3), (256, 14, 512, 1, 2, 0, 1)]
    B = 256
    for od in (torch.int32, torch.bfloat16):
        total_us, total_ops = 0.0, 0.0
        for C, H, Co, k, s, p, times in layers:
            x, w = rand_i8((B, C, H, H)), rand_i8((Co, C, k, k))
            alpha = torch.rand(Co, device="cuda") * 0.01
            Ho = (H + 2 * p - k) // s + 1
            outs = [torch.empty(B, Co, Ho, Ho, dtype=od, device="cuda") for _ in range(3)]
            cnt = [0]

            def fn():
                cnt[0] += 1
                N.conv2d_i8(x, w, s, p, alpha=alpha, out_dtype=od, out=outs[cnt[0] % 3])
            us = med_us(fn)
            ops = 2.0 * B * Co * Ho * Ho * C * k * k
            total_us += us * times
            total_ops += ops * times
            print(f"  conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s} x{times}: {us:7.1f} us  {ops / us / 1e6:6.0f} TOPS", flush=True)
            del x, w, outs
        print(f"config 3, ResNet-18 convolutions, batch {B}, out={str(od)[6:]}: {total_us:.0f} us in all, "
              f"{total_ops / total_us / 1e6:.0f} TOPS, {B / total_us * 1e6:.0f} images/s through the convolutions", flush=True)


    # the same stack as the reference's weight-only layers run it: floating-point activations (miq_conv2d_w8), beside
    # the library route (float convolution on a float copy of the weight + the fused epilogue kernel)
    import torch.nn.functional as F
    for dt in (torch.float32, torch.bfloat16):
        ours_us, lib_us = 0.0, 0.0
        for C, H, Co, k, s, p, times in layers:
            x = torch.randn(B, C, H, H, device="cuda").to(dt)
            w = rand_i8((Co, C, k, k))
            wf = w.to(dt)
            alpha = torch.rand(Co, device="cuda") * 0.01
            ours_us += times * med_us(lambda: N.conv2d_w8(x, w, s, p, alpha=alpha), iters=7)
            lib_us += times * med_us(lambda: N.epilogue_channels_f32(F.conv2d(x, wf, stride=s, padding=p).float().contiguous(),
                                                                      None, alpha, 0.0), iters=7)
            del x, w, wf
        print(f"config 3, weight-only, {str(dt)[6:]} activations, fp32 out: {ours_us:.0f} us in all ({B / ours_us * 1e6:.0f} images/s); "
              f"library convolution + epilogue {lib_us:.0f} us ({lib_us / ours_us:.2f}x)", flush=True)


def config4():
    torch.manual_seed(0)
    shapes = {"gate": (14336, 4096), "up": (14336, 4096), "down": (4096, 14336)}
    q = {}
    tq = 0.0
    for name, (n, k) in shapes.items():
        w = torch.randn(n, k, device="cuda") * 0.02
        tq += med_us(lambda: N.quantize_rows_sym_i8(w), iters=9)
        q[name] = N.quantize_rows_sym_i8(w)[:2]
        del w
    byts = sum(n * k * 5 + n * 12 for n, k in shapes.values())
    print(f"config 4, Llama-3-8B MLP: quantize gate+up+down {tq:.0f} us ({byts / tq / 1e3:.0f} GB/s)", flush=True)
    for M in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        xa, xd = rand_i8((M, 4096)), rand_i8((M, 14336))
        us = 0.0
        for name in ("gate", "up", "down"):
            wq, sc = q[name]
            x = xd if name == "down" else xa
            out = torch.empty(M, wq.shape[0], device="cuda")
            us += med_us(lambda: N.gemm_i8(x, wq, alpha=sc, out=out))
        ops = 2.0 * M * 3 * 14336 * 4096
        print(f"  M={M:3d}: gate+up+down {us:6.1f} us  {ops / us / 1e6:7.1f} TOPS  weights at {3 * 14336 * 4096 / us / 1e3:5.0f} GB/s", flush=True)
    # the same stack as the reference's weight-only layers run it: floating-point activations (miq_gemm_w8)
    for dt in (torch.bfloat16, torch.float32):
        for M in (1, 16, 64, 256):
            xa, xd = torch.randn(M, 4096, device="cuda").to(dt), torch.randn(M, 14336, device="cuda").to(dt)
            us = 0.0
            for name in ("gate", "up", "down"):
                wq, sc = q[name]
                x = xd if name == "down" else xa
                out = torch.empty(M, wq.shape[0], device="cuda")
                us += med_us(lambda: N.gemm_w8(x, wq, alpha=sc, out=out))
            ops = 2.0 * M * 3 * 14336 * 4096
            print(f"  weight-only, {str(dt).split('.')[-1]} activations, M={M:3d}: gate+up+down {us:6.1f} us  {ops / us / 1e6:7.1f} TFLOP/s  "
                  f"weights at {3 * 14336 * 4096 / us / 1e3:5.0f} GB/s", flush=True)


def config5():
    for n, k in ((28672, 8192), (8192, 28672), (8192, 8192)):
        w = rand_i8((n, k))
        sc = torch.rand(n, device="cuda") * 0.01
        for M in (256, 2048, 8192):
            x = rand_i8((M, k))
            out = torch.empty(M, n, device="cuda")
            us = med_us(lambda: N.gemm_i8(x, w, alpha=sc, out=out), iters=9)
            print(f"config 5, one GPU: M={M:4d} x [{n},{k}]: {us:7.1f} us  {2.0 * M * n * k / us / 1e6:6.0f} TOPS", flush=True)
            del x, out
        del w


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    if "3" in which:
        config3()
    if "4" in which:
        config4()
    if "5" in which:
        config5()
