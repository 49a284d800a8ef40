#!/usr/bin/env python3
"""The 3x3 / stride 1 patch kernels side by side: one workgroup per CU with the phases overlapped in software (the library's
choice) against two workgroups per CU (developer switch MIQ_CONV_PLANAR=2), on ResNet-18's 64->64 56x56 and 128->128 28x28
layers at batch 256; whole calls (for NCHW the re-layout launch included), median of 5 interleaved rounds, microseconds.

    python tools/conv_patch_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    spin @ spin
for C, H, Co in [(64, 56, 64), (128, 28, 128)]:
    x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Co, device="cuda") * 1e-4
    xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous()
    wp = N.conv2d_i8_pack_weight(w)
    for cl in (True, False):
        for od in ([torch.int8] if cl else []) + [torch.int32, torch.float32, torch.bfloat16]:
            if cl:
                outs = [torch.zeros(256, H + 2, H + 2, Co, dtype=od, device="cuda") for _ in range(3)]
                run = lambda i: N.conv2d_i8_nhwc(xp, wp, w.shape, 1, 1, alpha=alpha, out_dtype=od, out_pad=1, out=outs[i % 3])  # noqa: E731
            else:
                outs = [torch.empty(256, Co, H, H, dtype=od, device="cuda") for _ in range(3)]
                run = lambda i: N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])  # noqa: E731
            res = {}
            for _ in range(5):
                for v in ("1", "2"):
                    os.environ["MIQ_CONV_PLANAR"] = v
                    for i in range(3):
                        run(i)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(20):
                        run(i)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
            print(f"{C}->{Co} {H}x{H} {'channels-last' if cl else 'NCHW':13s} {str(od).split('.')[-1]:8s}: one workgroup per CU "
                  f"{statistics.median(res['1']):6.1f} | two {statistics.median(res['2']):6.1f}", flush=True)
