#!/usr/bin/env python3
"""SIZE^3 GEMM with the output rotating over several buffers (as a stack of layers writes it), A/B of
developer switches in one process:  python tools/gemm_rot.py SIZE ROUNDS LAUNCHES SPEC [SPEC ...]
SPEC = VARIANT[,ABL] as in tools/gemm_ab.py."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402
from tools.gemm_ab import set_variant  # noqa: E402

S, rounds, launches, specs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(-128, 128, (S, S), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
w = torch.randint(-128, 128, (S, S), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
alpha = torch.rand(S, generator=g, device="cuda") * 1e-3
for od in (torch.float32, torch.bfloat16):
    outs = [torch.empty(S, S, dtype=od, device="cuda") for _ in range(3)]
    times = {s: [] for s in specs}
    for _ in range(rounds):
        for s in specs:
            set_variant(s)
            for i in range(3):
                N.gemm_i8(a, w, alpha=alpha, out_dtype=od, out=outs[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                N.gemm_i8(a, w, alpha=alpha, out_dtype=od, out=outs[i % 3])
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / launches * 1e3)
    for s in specs:
        med = statistics.median(times[s])
        print(f"{str(od)[6:]} {s}: median {med:.1f} us ({2.0 * S ** 3 / med / 1e6:.0f} TOPS)  min {min(times[s]):.1f} us", flush=True)
