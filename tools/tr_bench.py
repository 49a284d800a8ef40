import sys, torch
sys.path.insert(0, '/root/repo')
from quantization_amd import _native as N
from quantization_amd import linear_w8a8
for shape in [(8192, 8192), (2048, 4096), (4096, 14336)]:
    a = torch.randint(-128, 128, shape, dtype=torch.int32, device='cuda').to(torch.int8)
    for _ in range(3): N.transpose_i8(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): N.transpose_i8(a)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(shape, f"{ms*1e3:.1f} us  {2*a.numel()/ms/1e6:.0f} GB/s")
x = torch.randint(-128, 127, (1024, 2048), dtype=torch.int32, device='cuda').to(torch.int8)
w = torch.randint(-128, 127, (2048, 4096), dtype=torch.int32, device='cuda').to(torch.int8)
for _ in range(3): linear_w8a8.linear_w8a8(x, w)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): linear_w8a8.linear_w8a8(x, w)
e.record(); torch.cuda.synchronize()
print("linear_w8a8 1024x2048x4096 (reference test shape):", f"{s.elapsed_time(e)/20*1e3:.1f} us")
