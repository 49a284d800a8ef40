#!/usr/bin/env python3
"""Llama-3-8B MLP decode step (gate, up: [14336,4096]; SiLU(gate) * up; down: [4096,14336]) on the weight-only kernels,
eager launches against one captured HIP graph replayed (the library allocates nothing and never synchronises, so
its launches capture as they are; workspaces are allocated once, before capture).

    python tools/graph_decode.py [M] [bf16|f32]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization_amd import _native as N  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dt = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
H, F = 4096, 14336
g = torch.Generator(device="cuda").manual_seed(0)
q = {}
for name, (n, k) in {"gate": (F, H), "up": (F, H), "down": (H, F)}.items():
    w = torch.randn(n, k, generator=g, device="cuda") * 0.02
    q[name] = N.quantize_rows_sym_i8(w)[:2]
    del w
x = torch.randn(M, H, generator=g, device="cuda").to(dt)
bufs = {"gate": torch.empty(M, F, device="cuda"), "up": torch.empty(M, F, device="cuda"), "down": torch.empty(M, H, device="cuda")}


def step():
    N.gemm_w8(x, q["gate"][0], alpha=q["gate"][1], out=bufs["gate"])
    N.gemm_w8(x, q["up"][0], alpha=q["up"][1], out=bufs["up"])
    h = (torch.nn.functional.silu(bufs["gate"]) * bufs["up"]).to(dt)
    N.gemm_w8(h, q["down"][0], alpha=q["down"][1], out=bufs["down"])
    return bufs["down"]


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


eager_out = step().clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(bufs["down"], eager_out), "graph replay differs from eager"
te, tg = timed(step), timed(graph.replay)
wbytes = 3 * H * F
print(f"MLP decode step, M={M}, {str(dt).split('.')[-1]} activations: eager {te:6.1f} us | HIP graph replay {tg:6.1f} us "
      f"(weights at {wbytes / tg / 1e3:5.0f} GB/s)", flush=True)
