"""The host side of the C ABI under -fsanitize=undefined (trap mode): tests/host_fuzz.py hands every entry point random and
nonsensical sizes in a child process; signed overflow, a division by zero or any other undefined step in the argument checks
and size arithmetic ends the child with SIGILL.  (What this found when it was written: a division by zero in two workspace
queries given K = 2^62, and size products that overflowed before their range checks in most entry points.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_entry_points_survive_nonsensical_sizes(seed):
    from quantization_amd import build
    if not os.path.exists(build.UBSAN_LIB):
        pytest.skip("libmiq_ubsan.so is not built (python -m quantization_amd.build --ubsan)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_fuzz.py"), build.UBSAN_LIB, str(seed), "150"],
                       capture_output=True, text=True, timeout=600)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1].startswith("DONE"), (
        f"child ended with {r.returncode} in or after: {lines[-1][:400] if lines else '(nothing printed)'}\n{r.stderr[-600:]}")
