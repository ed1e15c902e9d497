"""__graft_entry__.smoke() as the driver runs it (outside pytest: the library's own defaults, the count form among them -- tests/conftest.py
switches that off for the tests that pin the mask-form schedules).  Last in collection order on purpose: it repeats, end to end, what
the tests before it check piece by piece."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_as_the_driver_runs_it(native, monkeypatch):
    from jepsen_tigerbeetle_amd import core
    monkeypatch.setattr(core, "DEFAULT_COUNT_FORM", True)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
