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


@pytest.fixture
def monkeypatch():
    """pytest's monkeypatch, plus: the library reads its NB_* diagnostic overrides once per process (no launch path calls
    getenv), so setting / restoring one of them makes the library read them again (``nb_debug_reload_env``)."""
    from _pytest.monkeypatch import MonkeyPatch

    def reload():
        import nenbody_amd._lib as _lib

        if _lib._lib is not None:  # only once the library is loaded; a fresh load reads the environment itself
            _lib._lib.nb_debug_reload_env()

    class NbMonkeyPatch(MonkeyPatch):
        def setenv(self, name, value, prepend=None):
            super().setenv(name, value, prepend)
            if name.startswith("NB_"):
                reload()

        def delenv(self, name, raising=True):
            super().delenv(name, raising)
            if name.startswith("NB_"):
                reload()

        def undo(self):
            super().undo()
            reload()

    mp = NbMonkeyPatch()
    yield mp
    mp.undo()
