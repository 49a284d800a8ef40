import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N
def timed(fn, iters=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100): spin @ spin
for dt in (torch.bfloat16, torch.float32):
  for Nn, K in ((14336, 4096), (4096, 14336)):
    for M in (160, 192, 256, 384, 512):
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(Nn, device="cuda") * 0.01
        out = torch.empty(M, Nn, device="cuda")
        fn = lambda: N.gemm_w8(x, w, alpha=alpha, out=out)
        res = {}
        variants = [(None, None), (128, 128), (64, 256), (32, 256)] + ([(128, 256)] if dt == torch.bfloat16 else [])
        for _ in range(3):
            for v in variants:
                for k, val in (("MIQ_W8F_BM", v[0]), ("MIQ_W8F_BN", v[1])):
                    if val is None: os.environ.pop(k, None)
                    else: os.environ[k] = str(val)
                res.setdefault(v, []).append(timed(fn))
        print(f"{str(dt).split('.')[-1]:8s} M={M:4d} [{Nn},{K}]: " + " | ".join(f"{v[0]}x{v[1]} {sorted(t)[1]:6.1f}" for v, t in res.items()), flush=True)
