import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """Build libapi.so and the oracle pieces when they are missing (CPU box).  On the GPU
    box the prebuilt files travel with the snapshot."""
    lib = os.path.join(ROOT, "microimagelib_amd", "libapi.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, "microimagelib_amd", "bin", "reg3D")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "microimagelib_amd", "csrc"), "-j4"])
    ref = os.path.join(ROOT, "oracle", "_ref", "libpowell_ref.so")
    if not os.path.exists(ref) and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    yield


@pytest.fixture(scope="session")
def lib():
    from microimagelib_amd import _lib
    return _lib.load()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
