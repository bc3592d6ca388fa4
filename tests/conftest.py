import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The loaded libyacht_hip.so.  GPU tests FAIL (never skip) when it is missing: a GPU box
    that cannot load the HIP library must not look green."""
    from yacht_amd import _lib

    lib = _lib.load()
    assert _lib.device_count() >= 1, "no HIP device visible to libyacht_hip.so"
    return lib
