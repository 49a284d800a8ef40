"""Every entry point of the C ABI called with random and nonsensical sizes (zero, negative, 2^31, 2^40, 2^62, INT64_MAX
among plausible ones) on a library whose host side is built with -fsanitize=undefined in trap mode
(quantization_amd/libmiq_ubsan.so): an entry point must answer such a call with an error code or a zero size, without
arithmetic that overflows or divides by zero on the way.  Test infrastructure, run as a child process by
tests/test_host_sanitizer.py (undefined behaviour ends the child with SIGILL; the last line it printed names the call):

    python tests/host_fuzz.py LIB SEED ROUNDS

Pointers are null, so that a compute entry point refuses before it launches anything.  Where no GPU exists (the build
container) some are made-up non-null addresses instead: the checks then run through to the launch, which fails for want of a
device.  Never on a machine with a GPU: a made-up address would reach a kernel."""
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from quantization_amd import _native  # noqa: E402

_native.LIB_PATH = os.path.abspath(sys.argv[1])
seed, rounds = int(sys.argv[2]), int(sys.argv[3])
random.seed(seed)
L = _native.lib()
made_up = torch.cuda.device_count() == 0          # (counting devices does not initialise one)
plausible = 0.9 if seed % 2 == 0 else 0.5         # share of plausible sizes: deep into the checks / wide over the extremes
EXTREME = [0, 1, 2, 3, 4, 7, 16, 64, 255, 256, 4096, 65535, 65536, 1 << 20, (1 << 31) - 1, 1 << 31, (1 << 32) + 1, 1 << 40,
           1 << 62, (1 << 63) - 1, -1, -(1 << 62)]
SMALL = [0, 1, 2, 3, 5, 7, 8, 16, 28, 64, 100, 224, 512]
ADDRESSES = [0x10000000, 0x10000100, 0x10000010, 0x10000004, 0x10000002, 0x10000001]


def pointer():
    if not made_up or random.random() < 0.15:
        return None
    return random.choice(ADDRESSES)


calls = 0
for _ in range(rounds):
    for name, (res, args) in _native.SYMBOLS.items():
        if name in ("miq_last_error", "miq_version"):
            continue
        a = []
        for t in args:
            if t is ctypes.c_int64:
                a.append(random.choice(SMALL) if random.random() < plausible else random.choice(EXTREME))
            elif t is ctypes.c_int:
                a.append(random.choice([0, 1, 2, 3, 4, 5, 7, -1, 100, (1 << 31) - 1]))
            elif t is ctypes.c_size_t:
                a.append(random.choice([0, 1, 256, 1 << 20, (1 << 64) - 1]))
            elif t is ctypes.c_float:
                a.append(random.choice([0.0, 1.0, -1.5, float("inf"), float("nan")]))
            else:
                a.append(pointer())
        print(name, a, flush=True)
        getattr(L, name)(*a)
        calls += 1
print("DONE", calls, flush=True)
