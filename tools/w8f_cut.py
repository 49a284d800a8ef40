import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: E402,F401  (the developer build of the library: these switches exist only there)
from quantization_amd import _native as N
def timed(fn, iters=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
spin = torch.randn(4096, 4096, device="cuda")
for _ in range(100): spin @ spin
for dt in (torch.bfloat16, torch.float32):
    for M, Nn, K in [(2048, 8192, 8192), (2100, 8192, 8192), (2304, 8192, 8192), (2560, 8192, 8192), (4352, 8192, 8192)]:
        x = torch.randn(M, K, device="cuda").to(dt)
        w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device="cuda").to(torch.int8)
        alpha = torch.rand(Nn, device="cuda") * 0.01
        out = torch.empty(M, Nn, device="cuda")
        fn = lambda: N.gemm_w8(x, w, alpha=alpha, out=out)
        res = {}
        for _ in range(3):
            for v in ("1", "0"):
                os.environ["MIQ_W8F_ROWSPLIT"] = v
                res.setdefault(v, []).append(timed(fn))
        print(f"{str(dt).split('.')[-1]:8s} {M}x{Nn}x{K}: cut {sorted(res['1'])[1]:7.1f} us | uncut {sorted(res['0'])[1]:7.1f} us", flush=True)
