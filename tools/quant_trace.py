import sys, os, torch
sys.path.insert(0, os.getcwd())
from quantization_amd import _native as N
for rows, cols, dt in ((14336, 4096, torch.float32), (14336, 4096, torch.bfloat16), (4096, 4096, torch.float32), (4096, 14336, torch.float32), (28672, 8192, torch.float32)):
    esz = 4 if dt == torch.float32 else 2
    nbuf = max(4, int(1.0e9 // (rows * cols * esz)) + 1)
    ws = [(torch.randn(rows, cols, device="cuda") * 0.02).to(dt) for _ in range(nbuf)]
    for i in range(6 * nbuf):
        N.quantize_rows_sym_i8(ws[i % nbuf])
    torch.cuda.synchronize()
    del ws
