#!/usr/bin/env python3
"""What a pure write stream sustains: torch fills / copies over tensors far larger than the 256 MiB Infinity Cache, HIP events,
rotating buffers.  Calibrates the layers whose time is their output (ResNet-18's first convolution writes 822 MB)."""
import torch

dev = torch.device("cuda", 0)
bufs = [torch.empty(256 * 64 * 112 * 112, dtype=torch.float32, device=dev) for _ in range(3)]
src = torch.randn(256 * 64 * 112 * 112, dtype=torch.float32, device=dev)


def timed(fn, n=12):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


nb = bufs[0].numel() * 4
us = timed(lambda i: bufs[i % 3].fill_(1.5))
print(f"fill   {nb / 1e6:.0f} MB: {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s written")
us = timed(lambda i: bufs[i % 3].zero_())
print(f"zero   {nb / 1e6:.0f} MB: {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s written")
us = timed(lambda i: bufs[i % 3].copy_(src))
print(f"copy   {nb / 1e6:.0f} MB: {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s written (+ as much read)")
h = src[: src.numel() // 4]
us = timed(lambda i: torch.add(h, 1.0, out=bufs[i % 3][: h.numel()]))
print(f"add    {nb / 4e6:.0f} MB out: {us:7.1f} us  {nb / 4 / us / 1e6:5.2f} TB/s written (+ as much read)")
