import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


# PyTorch-ROCm bundles its own copies of libamdhip64 / libhsa-runtime64; libpbsim3_amd.so links the system ones.  Both
# can live in one process only if torch's copies are loaded FIRST (the other order leaves torch without a device:
# "no ROCm-capable device is detected").  Some GPU tests use torch after the product, so fix the load order here.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A test that waits for ever (ranks of a job that fell out of step, a kernel that spins) must fail, not hold the GPU box
    until the caller's limit: every test gets a ceiling (pytest-timeout; signal method, so a hanging subprocess.run is
    interrupted and its child killed)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for it in items:
        if not any(m.name == "timeout" for m in it.iter_markers()):
            it.add_marker(pytest.mark.timeout(300))
