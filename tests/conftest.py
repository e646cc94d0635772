import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available() -> bool:
    # Ask the HIP runtime through the product library itself (hipGetDeviceCount); torch is only a second
    # opinion -- its is_available() has been seen to answer False on a box whose GPU the library can use.
    try:
        from pixelbox_amd import capi

        if capi.device_count() > 0:
            return True
    except Exception:
        pass
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU would otherwise fail inside hipMalloc; skip loudly instead.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
