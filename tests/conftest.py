import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _mask_form_unless_asked():
    """The tests written before the count form pin the MASK-form schedules (one mask bit per crashed call) against their oracles
    (wgl_beam.c, wgl_window.c, sweep_ref.c): for them core.make_opts() leaves the count form off.  tests/test_count_form*.py ask for
    it by name (count_form=True), which is also what the library does by default."""
    from jepsen_tigerbeetle_amd import core
    old = core.DEFAULT_COUNT_FORM
    core.DEFAULT_COUNT_FORM = False
    yield
    core.DEFAULT_COUNT_FORM = old


@pytest.fixture(scope="session")
def native():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import jepsen_tigerbeetle_amd as pkg
    pkg.build()
    from jepsen_tigerbeetle_amd import _native
    _native.lib()
    return _native


@pytest.fixture(scope="session")
def oracle():
    from oracle import wgl
    wgl.build()
    return wgl


def has_gpu():
    try:
        from jepsen_tigerbeetle_amd import _native
        return _native.lib().tbc_device_count() > 0
    except Exception:
        return False
