"""Builds quantization_amd/libmiq.so (the HIP kernels + C ABI) for gfx950 with hipcc.

    python -m quantization_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is built in-tree next to this file so that it
travels with the source tree; it is never installed into site-packages.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmiq.so")
SOURCES = ["miq_api.hip", "miq_quantize.hip", "miq_gemm_i8.hip", "miq_gemm_w8f.hip", "miq_conv2d_i8.hip", "miq_pool.hip", "miq_activation.hip"]
HEADERS = ["miq_common.h", "miq_gemm_core.h"]

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


def build(force=False, verbose=False):
    """Compile each source to an object (in csrc/build/) and link libmiq.so.  Returns its path."""
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "miq.h"), __file__]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + common):
            cmd = [hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
