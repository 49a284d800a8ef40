#!/usr/bin/env python3
"""What bounds the 3x3 / stride 1 patch kernel: the developer library's timing ablations (MIQ_PLANAR_ABL bit 0 = no output
stores, bit 1 = no patch fetch, bit 2 = no MFMA loop: results are wrong by design; bit 3 = the MFMA's own lane placement in
the NCHW int32 / fp32 epilogue, results unchanged) on ResNet-18's 64->64 56x56 and 128->128 28x28 layers at batch 256, whole
calls (for NCHW the re-layout launch included), median of 5 interleaved rounds.

    python tools/conv_planar_abl.py [out: i32|f32|bf16|i8] [cl]      (cl = the channels-last entry point)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

od = {"i32": torch.int32, "f32": torch.float32, "bf16": torch.bfloat16, "i8": torch.int8}[sys.argv[1] if len(sys.argv) > 1 else "i32"]
cl = len(sys.argv) > 2 and sys.argv[2] == "cl"
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for C, H, Co in [(64, 56, 64), (128, 28, 128)]:
    x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Co, device="cuda") * 1e-4
    if cl:
        xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous()
        wp = N.conv2d_i8_pack_weight(w)
        outs = [torch.zeros(256, H + 2, H + 2, Co, dtype=od, device="cuda") for _ in range(3)]
        run = lambda i: N.conv2d_i8_nhwc(xp, wp, w.shape, 1, 1, alpha=alpha, out_dtype=od, out_pad=1, out=outs[i % 3])  # noqa: E731
    else:
        outs = [torch.empty(256, Co, H, H, dtype=od, device="cuda") for _ in range(3)]
        run = lambda i: N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])  # noqa: E731
    res = {}
    for _ in range(5):
        for v in ("0", "1", "2", "4", "3", "5", "6", "7") if cl else ("0", "8", "1", "6", "14", "7"):
            os.environ["MIQ_PLANAR_ABL"] = v
            for i in range(3):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{C}->{Co} {H}x{H}: " + " | ".join(f"abl {v} {statistics.median(t):6.1f}" for v, t in res.items()), flush=True)
