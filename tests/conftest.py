import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _reference_fp32_library():
    """The plain-fp32 (rocBLAS) formulation of the training step the MFMA kernels are checked against is a test-only
    helper library (tests/ref_fp32); make it known to the host layer for `train_precision="fp32"`."""
    path = os.path.join(ROOT, "tests", "ref_fp32", "libfsnerf_ref_fp32.so")
    try:
        import fs_nerf_amd  # noqa: F401
        from fs_nerf_amd import _lib
        _lib.register_reference_library(path)
    except Exception:
        pass
    yield
