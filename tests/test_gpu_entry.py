"""The driver's own entry points on the GPU box, run as it runs them."""
import pytest


@pytest.mark.gpu
def test_graft_entry_smoke_runs_to_its_end():
    """__graft_entry__.smoke(): the small invocation of the hot path the driver runs before the bench (it checks itself against
    the oracle and raises on any difference)"""
    import __graft_entry__ as g

    g.smoke()
