#!/usr/bin/env python3
"""A/B of the 3 x 3 planar conv kernel against the im2col-order kernel in ONE process, interleaved rounds.

    python tools/conv_ab.py B Cin H Cout [out: i32|f32|bf16] [rounds] [launches]

Whole calls (re-layout launch + conv kernel), outputs rotating over three buffers, HIP events around each
round of launches; prints the median and minimum per-call time over the rounds for both kernels."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

B, C, H, Co = (int(v) for v in sys.argv[1:5])
od = {"i32": torch.int32, "f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[5] if len(sys.argv) > 5 else "i32"]
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 7
launches = int(sys.argv[7]) if len(sys.argv) > 7 else 30
x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(Co, device="cuda") * 0.01
outs = [torch.empty(B, Co, H, H, dtype=od, device="cuda") for _ in range(3)]
ref = None
times = {"1": [], "0": []}
for r in range(rounds):
    for mode in ("1", "0"):
        os.environ["MIQ_CONV_PLANAR"] = mode
        for i in range(5):
            N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])
        if r == 0:
            if ref is None:
                ref = outs[0].clone()
            else:
                print("outputs equal:", torch.equal(ref, outs[0]), flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(launches):
            N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=outs[i % 3])
        e1.record()
        torch.cuda.synchronize()
        times[mode].append(e0.elapsed_time(e1) / launches * 1e3)
for mode, name in (("1", "planar"), ("0", "im2col-order")):
    t = times[mode]
    print(f"B={B} {C}->{Co} {H}x{H} {sys.argv[5] if len(sys.argv) > 5 else 'i32'} {name:13s}: median {statistics.median(t):7.1f} us  min {min(t):7.1f} us", flush=True)
