import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped, not failed, when collected without a device (plain `pytest`)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Read-only view of one tests/golden/*.npz fixture, grouped by case name."""

    def __init__(self, fname):
        self._z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)

    def cases(self, prefix=""):
        names = []
        for k in self._z.files:
            c = k.rsplit("/", 1)[0]
            if c.startswith(prefix) and c not in names:
                names.append(c)
        return names

    def get(self, case, field, default=None):
        key = f"{case}/{field}"
        return self._z[key] if key in self._z.files else default

    def raw(self, key):
        return self._z[key]

    def text(self, case, field):
        return bytes(self._z[f"{case}/{field}"]).decode()

    def has(self, case, field):
        return f"{case}/{field}" in self._z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(fname):
        if fname not in cache:
            cache[fname] = Golden(fname)
        return cache[fname]
    return load


def kind_of(case):
    """Storage precision of a quantizer fixture case, from its name."""
    if case.startswith("f16"):
        return "f16"
    if case.startswith("bf16"):
        return "bf16"
    return "f32"
