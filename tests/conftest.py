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

    def names_a_legacy_form(name, value):
        """the launch shapes only these knobs can name are not in the product library: the test that sets one
        runs against libnenbody_hip_legacy.so -- the same sources built with -DNB_LEGACY_FORMS -- until the patch is undone"""
        v = str(value)
        return ((name == "NB_STRICT_PC" and v != "0") or (name == "NB_STRICT_LANES" and v in ("2", "4", "8")) or name == "NB_STRICT_UNROLL"
                or (name == "NB_FAST_WAVES" and v != "8") or name == "NB_FAST_GROUPS" or name == "NB_FAST_PAIRS_W"
                or (name == "NB_STRICT_SL" and v in ("2", "3")))

    class NbMonkeyPatch(MonkeyPatch):
        _switched = False

        def setenv(self, name, value, prepend=None):
            super().setenv(name, value, prepend)
            if name.startswith("NB_"):
                if names_a_legacy_form(name, value) and not self._switched:
                    import nenbody_amd._lib as _lib

                    if not os.path.exists(_lib.LEGACY_LIB_PATH):
                        pytest.skip("libnenbody_hip_legacy.so is not built (make -C nenbody_amd/csrc legacy)")
                    _lib.use_library(_lib.LEGACY_LIB_PATH)
                    self._switched = True
                reload()

        def delenv(self, name, raising=True):
            super().delenv(name, raising)
            if name.startswith("NB_"):
                reload()

        def undo(self):
            super().undo()
            if self._switched:
                import nenbody_amd._lib as _lib

                _lib.use_library(None)
                self._switched = False
            reload()

    mp = NbMonkeyPatch()
    yield mp
    mp.undo()
