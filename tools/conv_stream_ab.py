#!/usr/bin/env python3
"""Cached against nontemporal output stores of the int8 convolution, interleaved rounds in one process (developer
library, MIQ_CONV_STREAM).  Whole calls with prepacked filters, outputs rotating over three buffers.

    python tools/conv_stream_ab.py [rounds] [launches]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
LAYERS = [(3, 64, 224, 7, 2, 3), (64, 64, 56, 3, 1, 1), (64, 128, 56, 3, 2, 1), (128, 128, 28, 3, 1, 1), (64, 128, 56, 1, 2, 0),
          (256, 256, 14, 3, 1, 1), (512, 512, 7, 3, 1, 1)]
B = 256
for cin, cout, hw, k, s, p in LAYERS:
    for od in (torch.int32, torch.float32):
        x = torch.randint(-128, 128, (B, cin, hw, hw), dtype=torch.int32, device="cuda").to(torch.int8)
        w = torch.randint(-128, 128, (cout, cin, k, k), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(cout, device="cuda") * 0.01
        packed = N.conv2d_i8_prepack(w)
        ho = (hw + 2 * p - k) // s + 1
        outs = [torch.empty(B, cout, ho, ho, dtype=od, device="cuda") for _ in range(3)]
        times = {"0": [], "1": []}
        ref = None
        for r in range(rounds):
            for mode in ("0", "1"):
                os.environ["MIQ_CONV_STREAM"] = mode
                for i in range(3):
                    N.conv2d_i8(x, w, s, p, alpha=alpha, out_dtype=od, out=outs[i % 3], packed=packed)
                if r == 0:
                    if ref is None:
                        ref = outs[0].clone()
                    else:
                        assert torch.equal(ref, outs[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(launches):
                    N.conv2d_i8(x, w, s, p, alpha=alpha, out_dtype=od, out=outs[i % 3], packed=packed)
                e1.record()
                torch.cuda.synchronize()
                times[mode].append(e0.elapsed_time(e1) / launches * 1e3)
        mb = outs[0].numel() * outs[0].element_size() / 1e6
        print(f"{cin:3d}->{cout:3d} {hw:3d}x{hw:<3d} k{k}s{s} {str(od)[6:]:8s} out {mb:6.0f} MB | cached {statistics.median(times['0']):7.1f} us | "
              f"nontemporal {statistics.median(times['1']):7.1f} us", flush=True)
        del outs, x
