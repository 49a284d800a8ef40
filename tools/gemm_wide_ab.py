#!/usr/bin/env python3
"""A/B of the 256 x 384 ("wide") GEMM tile against the 256 x 256 kernel in ONE process, interleaved rounds,
random data.  Needs the developer library (python -m quantization_amd.build --dev).

    python tools/gemm_wide_ab.py M N K ROUNDS LAUNCHES [OUT] SPEC [SPEC ...]

SPEC is WIDE[:STAGGER[:VARIANT]].  WIDE is the value of MIQ_GEMM_WIDE: 0 = never the wide tile, a count = that
many 384-channel tile columns, "auto" = the library's own choice; STAGGER = 1 selects the wide kernel's
one-barrier-per-phase form (MIQ_GEMM_WIDE_STAGGER), VARIANT the 256 x 256 kernel's schedule (MIQ_GEMM_VARIANT).
OUT: f32 (default), bf16 or i32.
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401
from quantization_amd import _native as N  # noqa: E402


def set_spec(spec):
    parts = (spec.split(":") + ["", ""])[:3]
    for name, val in zip(("MIQ_GEMM_WIDE", "MIQ_GEMM_WIDE_SCHED", "MIQ_GEMM_VARIANT"), parts):
        if val in ("", "auto"):
            os.environ.pop(name, None)
        else:
            os.environ[name] = val


def main():
    M, Nn, K, rounds, launches = (int(v) for v in sys.argv[1:6])
    rest = sys.argv[6:]
    od = {"f32": torch.float32, "bf16": torch.bfloat16, "i32": torch.int32}.get(rest[0]) if rest else None
    if od is not None:
        rest = rest[1:]
    else:
        od = torch.float32
    specs = rest
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Nn, generator=g, device="cuda") * 1e-3
    bias = torch.randn(Nn, generator=g, device="cuda")
    out = torch.empty(M, Nn, dtype=od, device="cuda")
    kw = {} if od is torch.int32 else {"alpha": alpha, "bias": bias}
    set_spec("0")
    ref = N.gemm_i8(a, w, out_dtype=od, **kw).clone()
    for s in specs:
        set_spec(s)
        c = N.gemm_i8(a, w, out_dtype=od, **kw)
        print(f"wide={s}: result {'equal' if torch.equal(c, ref) else 'DIFFERENT'}", flush=True)
    t_end = __import__("time").perf_counter() + 0.3
    while __import__("time").perf_counter() < t_end:   # clock ramp
        N.gemm_i8(a, w, out_dtype=od, out=out, **kw)
        torch.cuda.synchronize()
    times = {s: [] for s in specs}
    for _ in range(rounds):
        for s in specs:
            set_spec(s)
            for _ in range(3):
                N.gemm_i8(a, w, out_dtype=od, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                N.gemm_i8(a, w, out_dtype=od, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / launches * 1e3)
    ops = 2.0 * M * Nn * K
    for s in specs:
        med, mn = statistics.median(times[s]), min(times[s])
        print(f"wide={s}: median {med:.1f} us ({ops/med/1e6:.0f} TOPS)  min {mn:.1f} us ({ops/mn/1e6:.0f} TOPS)", flush=True)


if __name__ == "__main__":
    main()
