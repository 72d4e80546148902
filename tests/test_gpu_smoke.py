"""GPU: the driver's round-end smoke (`__graft_entry__.smoke()`) as a test, so a change that breaks it shows up in `pytest -m gpu`."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke():
    import __graft_entry__ as entry
    entry.smoke()
