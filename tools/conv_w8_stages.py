#!/usr/bin/env python3
"""Tile sweep of miq_conv2d_w8 on ResNet-18 layers at batch 256 (developer switch MIQ_CONV_W8_TILE: a = 128x64,
b = 128x128, c = 64x64, d = 64x128, e = 64x64 deepest ring, f = 64x256, g = 128x256 one plane only, 0 = the library's own
choice): whole calls, median of 3 interleaved rounds, bf16 then fp32 activations.

    python tools/conv_w8_stages.py [variants, default 0abcde] [index of the first layer of the list]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N
def timed(fn, iters=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100): spin @ spin
for dt in (torch.bfloat16, torch.float32):
    for B, C, H, Co, k, s, p in [(256, 64, 56, 64, 3, 1, 1), (256, 64, 56, 128, 3, 2, 1), (256, 128, 28, 128, 3, 1, 1), (256, 128, 28, 256, 3, 2, 1), (256, 256, 14, 256, 3, 1, 1), (256, 256, 14, 512, 3, 2, 1), (256, 512, 7, 512, 3, 1, 1), (256, 3, 224, 64, 7, 2, 3)][int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
        x = torch.randn(B, C, H, H, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(Co, device="cuda") * 0.01
        fn = lambda: N.conv2d_w8(x, w, s, p, alpha=alpha)
        res = {}
        for _ in range(3):
            for v in (sys.argv[1] if len(sys.argv) > 1 else "0abcde"):
                os.environ["MIQ_CONV_W8_TILE"] = v
                res.setdefault(v, []).append(timed(fn))
        print(f"{str(dt).split('.')[-1]:8s} {C}->{Co} {H}x{H} k{k}s{s}: default " + " | ".join(f"{v} {sorted(t)[1]:6.1f}" for v, t in res.items()), flush=True)
