import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--slow", action="store_true", default=False, help="also run the cases marked slow (second parametrisations of "
                     "the long GPU tests, the extra bench subprocesses): tools/final_round.sh passes it, the default -m gpu run does not")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a second parametrisation of a long GPU test; skipped unless --slow or XWB_SLOW=1 "
                                       "(keeps the default -m gpu run well inside the driver's limit; one representative of each stays)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--slow") or os.environ.get("XWB_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow case: run with --slow (tools/final_round.sh does)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


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
