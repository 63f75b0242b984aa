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


import pytest


@pytest.fixture(autouse=True)
def _reference_elimination_order():
    """The stored fixtures (whole interior-point runs, residual histories, Phase-I cases) were generated with the reference's
    ordering sequence -- maximum cardinality search, then minimum degree (solvers.py:301-308) -- and are compared to 1e-9: the
    suite pins it.  The drivers' own default (options['peo'] = 'auto': the given order when it has zero fill) is covered by
    test_natural_elimination_order_gives_the_same_optimum."""
    from smcp_amd import solvers
    old = solvers.options.get("peo", "auto")
    solvers.options["peo"] = "mcs"
    yield
    solvers.options["peo"] = old
