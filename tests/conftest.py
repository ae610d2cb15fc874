import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip():
    """Loaded HIP library + torch; GPU tests fail (not skip) if the native path is missing."""
    import torch
    from gadfly_amd import _lib
    assert torch.cuda.is_available(), "GPU test collected without a HIP device"
    _lib.load()
    return _lib
