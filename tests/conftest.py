import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/nbody_oracle.c."""
    import oracle as _oracle

    _oracle.build()
    _oracle.load()
    return _oracle


@pytest.fixture(scope="session")
def nb():
    """The product package; the HIP library must already be built (python -c 'import __graft_entry__ as g; g.build()')."""
    import nenbody_amd

    nenbody_amd.load()
    return nenbody_amd


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
