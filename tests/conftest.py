import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """libvitmi.so, built on demand (hipcc cross-compiles without a GPU)."""
    from vit_torch_amd import _lib, build
    if not _lib.LIB_PATH.exists():
        build.build()
    return _lib.load()
