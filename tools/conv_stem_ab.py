#!/usr/bin/env python3
"""ResNet-18's first layer (3 -> 64, 7 x 7 / 2, 224 x 224, batch 256) on the stem kernel against the im2col-order kernel
(developer switch MIQ_CONV_STEM=0): whole calls of the channels-last entry point, median of 5 interleaved rounds,
microseconds.  (int32 outputs and the NCHW entry point stay on the im2col-order kernel, measured faster there.)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

B, C, H, Co = 256, 3, 224, 64
x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (Co, C, 7, 7), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Co, device="cuda") * 1e-4
xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 1, 3, 3, 3, 3)).contiguous()
w4 = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 1)).contiguous()
wp = N.conv2d_i8_pack_weight(w4)
for cl in (True,):
    for od in [torch.int8, torch.float32, torch.bfloat16]:
        if cl:
            outs = [torch.zeros(B, 114, 114, Co, dtype=od, device="cuda") for _ in range(2)]
            run = lambda i: N.conv2d_i8_nhwc(xp, wp, w4.shape, 2, 3, alpha=alpha, out_dtype=od, out_pad=1, out=outs[i % 2])  # noqa: E731
        else:
            outs = [torch.empty(B, Co, 112, 112, dtype=od, device="cuda") for _ in range(2)]
            run = lambda i: N.conv2d_i8(x, w, 2, 3, alpha=alpha, out_dtype=od, out=outs[i % 2])  # noqa: E731
        res = {}
        for _ in range(5):
            for v in ("1", "0"):
                os.environ["MIQ_CONV_STEM"] = v
                for i in range(2):
                    run(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(10):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"{'channels-last' if cl else 'NCHW':13s} {str(od).split('.')[-1]:8s}: stem kernel {statistics.median(res['1']):6.1f} | "
              f"im2col-order kernel {statistics.median(res['0']):6.1f}", flush=True)
