import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    _oracle.lib()
    return _oracle


@pytest.fixture(params=["libm", "xwb_trig"])
def trig(request, oracle):
    """cos / sin of the oracle's SimpleRace and goal-warp call sites: the host's libm (the oracle's default: a checker that
    shares no arithmetic with the HIP kernels) or include/xwb_trig.h (the kernels' own definition).  Tests that take this
    fixture run against both; everything else runs against libm."""
    L = oracle.lib()
    L.orc_set_trig_libm(1 if request.param == "libm" else 0)
    yield request.param
    L.orc_set_trig_libm(1)
