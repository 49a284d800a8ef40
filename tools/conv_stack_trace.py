#!/usr/bin/env python3
"""ResNet-18's eleven distinct convolutions at batch 256 with prepacked filters, each called a few times, for a
rocprofv3 --kernel-trace run (tools/trace_medians.py prints the per-kernel medians in launch order):
    python tools/conv_stack_trace.py [i8|f32|bf16] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import RESNET18_CONVS  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "i8"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
B = 256
for C, H, Co, k, s, p, times in RESNET18_CONVS:
    w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    Ho = (H + 2 * p - k) // s + 1
    if kind == "i8":
        x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
        outs = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
        pk = N.conv2d_i8_pack_weight(w)
        for i in range(reps):
            N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[i % 3], packed=pk)
    else:
        dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[kind]
        x = torch.randn(B, C, H, H, device=dev).to(dt)
        alpha = torch.rand(Co, device=dev) * 1e-3
        outs = [torch.empty(B, Co, Ho, Ho, device=dev) for _ in range(3)]
        pk = N.conv2d_w8_pack_weight(w, dt)
        for i in range(reps):
            N.conv2d_w8(x, w, s, p, alpha=alpha, out=outs[i % 3], packed=pk)
    torch.cuda.synchronize()
    print(f"{C}->{Co} {H}x{H} k{k}s{s} done", flush=True)
    del x, w, outs
