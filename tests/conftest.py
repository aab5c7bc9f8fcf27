import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libref_shared.so (built only where /root/reference exists)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """torch ships its own HIP runtime; libhalo_hip.so links the system one.  Both live in one process in the tests that
    use torch.distributed plumbing (as in bench.py, where torch initialises first): bring torch's runtime up before the
    first backend is created, so late initialisation cannot find the device already claimed."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
