import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the in-tree libraries once (hipcc cross-compiles without a GPU)."""
    from smcp_amd import build as b
    b.build(verbose=False)
    from oracle import oracle
    oracle.build()
