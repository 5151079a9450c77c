import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PIXEL_CASES = ["perf_2p0x", "balanced_1p7x", "quality_1p5x", "ultra_1p3x", "odd_2p0x_ragged"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are skipped (not failed) on a box without an MI355X, so a plain `pytest tests/` is green there."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def fsr():
    """The product package (fidelityfx-fsr_amd), built if necessary."""
    pkg = importlib.import_module("fidelityfx-fsr_amd")
    pkg.build()
    pkg.load()
    return pkg


@pytest.fixture(scope="session")
def port():
    import cpu_oracle
    return cpu_oracle.port()


@pytest.fixture(scope="session")
def ref():
    import cpu_oracle
    if not cpu_oracle.have_ref() and not os.path.exists("/root/reference/ffx-fsr/ffx_fsr1.h"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return cpu_oracle.ref()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """float32 arrays bit-identical, treating any NaN as equal to any NaN."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))
