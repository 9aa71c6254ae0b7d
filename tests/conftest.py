import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def built():
    """Compile the native pieces once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    from edipack_amd import capi
    if capi.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    capi.init(0)
    return True
