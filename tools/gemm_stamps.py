#!/usr/bin/env python3
"""Where the 256x256 GEMM kernel's wave cycles go: runs the timing-only build that stamps
s_memtime between the segments of every phase and prints cycles per K-tile for each
(phase, segment), median over waves, for the two wave groups.

    python tools/gemm_stamps.py [size] [s|t]      (t: the two-segment schedule)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(size, generator=g, device="cuda") * 1e-3
out = torch.empty(size, size, dtype=torch.float32, device="cuda")
tiles = ((size + 255) // 256) ** 2
stamps = torch.zeros(tiles * 8 * 20, dtype=torch.int32, device="cuda")
which = sys.argv[2] if len(sys.argv) > 2 else "s"
os.environ["MIQ_GEMM_ABL"] = which
os.environ["MIQ_GEMM_STAMPS"] = f"{stamps.data_ptr():x}"
for _ in range(20):
    N.gemm_i8(a, w, alpha=alpha, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    N.gemm_i8(a, w, alpha=alpha, out=out)
e1.record()
torch.cuda.synchronize()
print(f"stamped build: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
s = stamps.view(tiles, 8, 4, 5).float() / (size / 128)   # cycles per K-tile
names = ["issue reads+DMA", "wait DMA", "barrier 1", "MFMA issue", "barrier 2"]
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    med = s[:, sl].reshape(-1, 4, 5).median(dim=0).values
    print(grp)
    for p in range(4 if which == "s" else 2):
        print(f"  phase {p + 1}: " + "  ".join(f"{names[i]} {med[p, i]:6.0f}" for i in range(5)) + f"   total {med[p].sum():6.0f}")
    print(f"  K-tile total {med.sum():.0f} cycles; by segment: " + "  ".join(f"{names[i]} {med[:, i].sum():.0f}" for i in range(5)))
