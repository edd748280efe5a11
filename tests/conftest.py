import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A hung rendezvous of a two-process test (a port still in TIME_WAIT) must fail that test, not stall the whole run:
    every test gets a 20-minute ceiling (pytest-timeout; the process groups themselves time out after 5 minutes)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(1200))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["default", "g8_everywhere"])
def g8(request):
    """The 16-bit conv op tests under both dispatch modes of the 8-wave kernel of the wide layers (csrc/conv_cl16_g8.hip):
    the default (it takes only launches that fill the chip: none of the small test shapes) and slv_cl16_g8_mode(2) = every
    launch it can express.  Plans carry the statistics-partial count of the kernel that runs: dropped on both sides."""
    from selavi_amd import ops16
    from selavi_amd._lib import C
    prev = C.slv_cl16_g8_mode(2 if request.param == "g8_everywhere" else -1)
    ops16.Plan16._cache.clear()
    yield request.param
    C.slv_cl16_g8_mode(prev)
    ops16.Plan16._cache.clear()
