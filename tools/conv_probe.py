#!/usr/bin/env python3
"""Run one conv layer repeatedly (for rocprofv3 kernel traces and counter passes).

    python tools/conv_probe.py B Cin H Cout k stride pad [iters] [out: i32|f32|bf16|i8] [cl]
(cl = the channels-last entry point, output with a border of 1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

B, C, H, Co, k, s, p = (int(v) for v in sys.argv[1:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 50
od = {"i32": torch.int32, "f32": torch.float32, "bf16": torch.bfloat16, "i8": torch.int8}[sys.argv[9] if len(sys.argv) > 9 else "i32"]
cl = len(sys.argv) > 10 and sys.argv[10] == "cl"
x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Co, device="cuda") * 0.01
Ho = (H + 2 * p - k) // s + 1
if cl:
    cpad = (-C) % 4   # the channels-last entry point takes whole 4-byte pixels
    xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, cpad, p, p, p, p)).contiguous()
    w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cpad)).contiguous()
    wp = N.conv2d_i8_pack_weight(w)
    outs = [torch.zeros(B, Ho + 2, Ho + 2, Co, dtype=od, device="cuda") for _ in range(3)]
    for i in range(iters):
        N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, alpha=alpha, out_dtype=od, out_pad=1, out=outs[i % 3])
else:
    outs = [torch.empty(B, Co, Ho, Ho, dtype=od, device="cuda") for _ in range(3)]
    for i in range(iters):
        N.conv2d_i8(x, w, s, p, alpha=alpha, out_dtype=od, out=outs[i % 3])
torch.cuda.synchronize()
print("done", flush=True)
