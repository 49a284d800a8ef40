#!/usr/bin/env python3
"""Where a workgroup of the 3x3 / stride 1 patch kernel spends a stage: shader-clock stamps of the developer library
(MIQ_PLANAR_STAMPS) on ResNet-18's 64->64 56x56 and 128->128 28x28 layers at batch 256; medians over the workgroups, in
cycles: wait for the stage's fetch | barrier | decode of the next stage | the nine taps (with the next fetch's pieces and,
where a unit has just finished, its rows leaving among them).

    python tools/conv_planar_stamps.py [out: i32|f32|bf16|i8] [cl|nchw] [Cin,H,Cout ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

od = {"i32": torch.int32, "f32": torch.float32, "bf16": torch.bfloat16, "i8": torch.int8}[sys.argv[1] if len(sys.argv) > 1 else "i32"]
cl = len(sys.argv) > 2 and sys.argv[2] == "cl"
stamps = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
LAYERS = [(64, 56, 64), (128, 28, 128)]
if len(sys.argv) > 3:   # e.g. 256,14,256
    LAYERS = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
for C, H, Co in LAYERS:
    x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Co, device="cuda") * 1e-4
    if cl:
        xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous()
        wp = N.conv2d_i8_pack_weight(w)
        out = torch.zeros(256, H + 2, H + 2, Co, dtype=od, device="cuda")
        run = lambda: N.conv2d_i8_nhwc(xp, wp, w.shape, 1, 1, alpha=alpha, out_dtype=od, out_pad=1, out=out)  # noqa: E731
    else:
        out = torch.empty(256, Co, H, H, dtype=od, device="cuda")
        run = lambda: N.conv2d_i8(x, w, 1, 1, alpha=alpha, out_dtype=od, out=out)  # noqa: E731
    for _ in range(20):
        run()
    os.environ["MIQ_PLANAR_STAMPS"] = str(stamps.data_ptr())
    stamps.zero_()
    run()
    torch.cuda.synchronize()
    del os.environ["MIQ_PLANAR_STAMPS"]
    t = stamps.view(512, 8, 8).cpu()
    names = ["wait", "barrier", "next", "taps", "-", "-"]
    print(f"{C}->{Co} {H}x{H} ({'channels-last' if cl else 'NCHW'}, {sys.argv[1] if len(sys.argv) > 1 else 'i32'} out)")
    for it in range(8):
        ok = t[:, it, 6] > 0
        if int(ok.sum()) < 16:
            break
        d = (t[ok, it, 1:7] - t[ok, it, 0:6]).double()
        tot = (t[ok, it, 6] - t[ok, it, 0]).double()
        print(f"  stage {it} ({int(ok.sum())} workgroups): " + " | ".join(f"{n} {int(d[:, k].median()):6d}" for k, n in enumerate(names)) +
              f" | whole {int(tot.median()):6d}")
    ok = t[:, 0, 0] > 0
    life = []
    for wg in range(256):
        last = max(int(t[wg, it, 6]) for it in range(8))
        if last:
            life.append(last - int(t[wg, 0, 0]))
    print(f"  first stamp to last stamp per workgroup: median {sorted(life)[len(life) // 2]} cycles")
