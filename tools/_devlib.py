"""Points quantization_amd._native at libmiq_dev.so, the build that honours the developer switches
(MIQ_GEMM_*, MIQ_CONV_*, ...) and holds the timing-only kernels.  Import this before the first call
into _native:

    import _devlib  # noqa: F401

Build it with `python -m quantization_amd.build --dev` (in the build container: GPU boxes only run).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from quantization_amd import _native  # noqa: E402
from quantization_amd import build as _build  # noqa: E402

if not os.path.exists(_build.DEV_LIB):
    raise SystemExit(f"{_build.DEV_LIB} is missing: run `python -m quantization_amd.build --dev` first")
assert _native._lib is None, "import _devlib before the first call into quantization_amd._native"
_native.LIB_PATH = _build.DEV_LIB
