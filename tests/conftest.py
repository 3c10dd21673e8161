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


@pytest.fixture(autouse=True)
def _debug_switches_back_to_defaults():
    """The library's vitmi_debug_* switches are process-wide: whatever a test flips (and however it ends — the
    part after `yield` runs on failures too) is undone before the next test starts."""
    yield
    from vit_torch_amd import _lib
    # reset the library this process actually uses (the VITMI_LIB override included), and only if a test loaded it:
    # a CPU-only run never dlopens the HIP .so from here
    lib_ = getattr(_lib, "_lib", None)
    if lib_ is not None:
        try:
            lib_.vitmi_debug_reset()
        except (OSError, AttributeError):
            pass
