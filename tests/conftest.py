import os
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # the test process owns its environment (pcabo/_native.py: Batch() warns otherwise)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "para-ortho-pca-bo_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def native():
    """The ctypes binding; building the library first if it is missing (build is CPU-only)."""
    lib = os.path.join(PKG, "lib", "libpcabo.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    from pcabo import _native
    return _native
