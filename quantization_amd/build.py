"""Builds quantization_amd/libmiq.so (the HIP kernels + C ABI) for gfx950 with hipcc.

    python -m quantization_amd.build [--force] [--dev] [--ubsan]

hipcc cross-compiles without a GPU.  The .so is built in-tree next to this file so that it
travels with the source tree; it is never installed into site-packages.

libmiq.so, the product, is compiled with -DMIQ_NO_DEV_SWITCHES: it reads no environment variable
and holds no timing-only kernel.  --dev also builds libmiq_dev.so from the same sources without
that define: the developer A/B switches (MIQ_GEMM_*, MIQ_CONV_*, ...) and the ablation builds of
the GEMM kernel, which tools/ load through tools/_devlib.py.  Nothing in the package loads it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmiq.so")
DEV_LIB = os.path.join(HERE, "libmiq_dev.so")
SOURCES = ["miq_api.hip", "miq_quantize.hip", "miq_gemm_i8.hip", "miq_gemm_w8f.hip", "miq_conv2d_i8.hip", "miq_conv2d_general.hip", "miq_pool.hip", "miq_activation.hip"]
HEADERS = ["miq_common.h", "miq_gemm_core.h", "miq_patch.h", "miq_conv_dense.h"]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # The quantizer's results depend on IEEE division and on unfused multiply/add.
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-ffp-contract=off",
    "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
    "-I", os.path.join(ROOT, "include"), "-I", CSRC,
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# The product's sources with the HOST side compiled under -fsanitize=undefined in trap mode (no sanitizer runtime to load:
# undefined behaviour ends the process with SIGILL): what tests/test_host_sanitizer.py hands nonsensical sizes to.  Device
# code is compiled as ever; GPU sanitizers are not used.
UBSAN_LIB = os.path.join(HERE, "libmiq_ubsan.so")
UBSAN_FLAGS = ["-Xarch_host", "-fsanitize=undefined,integer-divide-by-zero", "-Xarch_host", "-fsanitize-trap=all"]


def build(force=False, verbose=False, dev=False, ubsan=False):
    """Compile each source to an object (in csrc/build/) and link libmiq.so (dev=True: libmiq_dev.so, with
    the developer switches; ubsan=True: libmiq_ubsan.so, the product with a sanitized host side).  Returns the library's path."""
    objdir = os.path.join(CSRC, "build", "ubsan" if ubsan else "dev" if dev else "product")
    os.makedirs(objdir, exist_ok=True)
    lib = UBSAN_LIB if ubsan else DEV_LIB if dev else LIB
    flags = FLAGS + ([] if dev else ["-DMIQ_NO_DEV_SWITCHES"]) + (UBSAN_FLAGS if ubsan else [])
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "miq.h"), __file__]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + common):
            cmd = [hipcc()] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _stale(lib, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
    if "--ubsan" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, ubsan=True))
