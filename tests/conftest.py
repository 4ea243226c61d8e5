import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree artefacts exist (libnfcgpu.so, hostsim, oracle)."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    return True
