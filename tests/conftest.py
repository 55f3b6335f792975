import os
import sys

import pytest

try:  # one HIP runtime per process: PyTorch's bundled copy must be the first one loaded
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference and oracle/_ref (authoring container only)")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """Build the CPU oracle once per session (a few seconds of gcc)."""
    import _harness

    _harness.build_oracle()
