import torch, torch.nn.functional as F, time
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for dt in (torch.float32, torch.bfloat16):
    for (B, C, H, Co, k, s, p) in [(256, 64, 56, 64, 3, 1, 1), (256, 128, 28, 128, 3, 1, 1), (256, 256, 14, 256, 3, 1, 1), (256, 512, 7, 512, 3, 1, 1), (256, 3, 224, 64, 7, 2, 3), (64, 3, 32, 64, 3, 1, 1), (64, 512, 4, 512, 3, 1, 1)]:
        x = torch.randn(B, C, H, H, device="cuda", dtype=dt)
        w = torch.randint(-128, 128, (Co, C, k, k), device="cuda").to(dt)
        t = timed(lambda: F.conv2d(x, w, stride=s, padding=p))
        Ho = (H + 2 * p - k) // s + 1
        fl = 2.0 * B * Co * Ho * Ho * C * k * k
        print(f"{str(dt).split('.')[-1]:8s} B={B} {C}->{Co} {H}x{H} k{k}s{s}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
