#!/usr/bin/env python3
"""A/B of the subsampling re-layout for strided 1 x 1 convolutions against re-laying the whole image, in ONE
process, interleaved rounds (developer switch MIQ_CONV_SUBSAMPLE).

    python tools/conv_sub_ab.py [out: i32|f32|bf16] [rounds] [launches]

ResNet-18's three downsample layers at batch 256, int8 activations (miq_conv2d_i8) and, with `w8`, fp32 / bf16
activations (miq_conv2d_w8).  Whole calls, outputs rotating over three buffers, HIP events around each round."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

od_name = sys.argv[1] if len(sys.argv) > 1 else "i32"
od = {"i32": torch.int32, "f32": torch.float32, "bf16": torch.bfloat16}[od_name]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 30
SHAPES = [(256, 64, 56, 128), (256, 128, 28, 256), (256, 256, 14, 512)]


def ab(name, call, outs):
    ref = None
    times = {"1": [], "0": []}
    for r in range(rounds):
        for mode in ("1", "0"):
            os.environ["MIQ_CONV_SUBSAMPLE"] = mode
            for i in range(5):
                call(outs[i % 3])
            if r == 0:
                if ref is None:
                    ref = outs[0].clone()
                else:
                    print("outputs equal:", torch.equal(ref, outs[0]), flush=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                call(outs[i % 3])
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / launches * 1e3)
    for mode, label in (("1", "subsampled"), ("0", "whole image")):
        t = times[mode]
        print(f"{name} {label:11s}: median {statistics.median(t):7.1f} us  min {min(t):7.1f} us", flush=True)


for B, C, H, Co in SHAPES:
    w = torch.randint(-128, 128, (Co, C, 1, 1), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Co, device="cuda") * 0.01
    Ho = (H - 1) // 2 + 1
    x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    outs = [torch.empty(B, Co, Ho, Ho, dtype=od, device="cuda") for _ in range(3)]
    ab(f"int8 B={B} {C}->{Co} {H}x{H} k1s2 {od_name}",
       lambda o: N.conv2d_i8(x, w, 2, 0, alpha=alpha, out_dtype=od, out=o), outs)
    if "w8" in sys.argv and hasattr(N, "conv2d_w8"):
        for xd in (torch.float32, torch.bfloat16):
            xf = torch.randn(B, C, H, H, device="cuda", dtype=xd)
            fo = [torch.empty(B, Co, Ho, Ho, dtype=torch.float32, device="cuda") for _ in range(3)]
            ab(f"w8 {str(xd)[6:]} B={B} {C}->{Co} {H}x{H} k1s2",
               lambda o: N.conv2d_w8(xf, w, 2, 0, alpha=alpha, out=o), fo)
