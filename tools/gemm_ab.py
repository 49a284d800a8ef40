#!/usr/bin/env python3
"""A/B of the GEMM kernel's developer variants in ONE process, interleaved rounds, random data.

    python tools/gemm_ab.py SIZE ROUNDS LAUNCHES VARIANT[,ABL] [VARIANT[,ABL] ...]

Every variant's int32 result is first compared with variant 0's (bit-equal, unless an ablation
switch is set), then ROUNDS rounds of LAUNCHES launches each are timed with HIP events; the median
and the minimum over rounds are printed per variant.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N  # noqa: E402


def set_variant(spec):
    var, _, abl = spec.partition(",")
    os.environ["MIQ_GEMM_VARIANT"] = var
    if abl:
        os.environ["MIQ_GEMM_ABL"] = abl
    else:
        os.environ.pop("MIQ_GEMM_ABL", None)


def main():
    size, rounds, launches = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    specs = sys.argv[4:]
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (size, size), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(size, generator=g, device="cuda") * 1e-3
    out = torch.empty(size, size, dtype=torch.float32, device="cuda")
    set_variant("0")
    ref = N.gemm_i8(a, w, out_dtype=torch.int32)
    for s in specs:
        set_variant(s)
        c = N.gemm_i8(a, w, out_dtype=torch.int32)
        print(f"variant {s}: int32 result {'equal' if torch.equal(c, ref) else 'DIFFERENT'}", flush=True)
    times = {s: [] for s in specs}
    for _ in range(rounds):
        for s in specs:
            set_variant(s)
            for _ in range(3):
                N.gemm_i8(a, w, alpha=alpha, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                N.gemm_i8(a, w, alpha=alpha, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / launches * 1e3)
    for s in specs:
        med, mn = statistics.median(times[s]), min(times[s])
        ops = 2.0 * size ** 3
        print(f"variant {s}: median {med:.1f} us ({ops/med/1e6:.0f} TOPS)  min {mn:.1f} us ({ops/mn/1e6:.0f} TOPS)", flush=True)


if __name__ == "__main__":
    main()
