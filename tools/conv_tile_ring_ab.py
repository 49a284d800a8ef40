#!/usr/bin/env python3
"""Tile shapes and ring depths of the int8 convolution's im2col-order kernel (developer switch MIQ_CONV_TILE) on the
ResNet-18 layers that run on it, batch 256, whole calls as replayed HIP graphs, interleaved in one process; every variant's
output is compared with the default's.    python tools/conv_tile_ring_ab.py [letters, default "-edEFDGHIJ"]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
import bench  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

letters = sys.argv[1] if len(sys.argv) > 1 else "-edEFDGHIJ"
B = 256
dev = torch.device("cuda", 0)
for C, H, Co, k, s, p, times in bench.RESNET18_CONVS:
    if C < 64 or (k == 3 and s == 1 and C < 512):
        continue
    x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
    w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
    Ho = (H + 2 * p - k) // s + 1
    outs = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
    pk = N.conv2d_i8_prepack(w, (B, C, H, H), s, p)
    it = [0]

    def step():
        N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[it[0] % 3], packed=pk)
        it[0] += 1
    os.environ.pop("MIQ_CONV_TILE", None)
    os.environ["MIQ_CONV_PLANAR"] = "0" if C == 512 else "1"
    step()
    ref = outs[(it[0] - 1) % 3].clone()
    line = []
    for rnd in range(2):
        for t in letters:
            if t == "-":
                os.environ.pop("MIQ_CONV_TILE", None)
            else:
                os.environ["MIQ_CONV_TILE"] = t
            step()
            ok = torch.equal(outs[(it[0] - 1) % 3], ref)
            ms = bench.graphed_ms(step, 6, 10, 12)
            if rnd == 1:
                line.append(f"{t} {ms * 1e3:5.1f}{'' if ok else ' WRONG'}")
    print(f"conv {C:3d}->{Co:3d} {H:3d}x{H:<3d} k{k}s{s}: " + "  ".join(line) + " us", flush=True)
