import os
import sys

import pytest

# The GPU tests force code paths at toy sizes (cluster sizes, re-planning marks, the XCD placement self-test, the broker's
# tick mode) with DEVELOPMENT knobs, which the library reads only behind JD_DEV=1 (csrc/jd_internal.h: jd_dev_env).  What a
# caller needs - the batch pipeline, capacities, partial traces - goes through the C ABI in these tests, as in bench.py.
os.environ.setdefault("JD_DEV", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs must fail, not take the whole run with it (pytest-timeout, where it is installed)
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if "gpu" in it.keywords and not it.get_closest_marker("timeout"):
                it.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def built():
    """Build the HIP extension and the oracle once per session."""
    from juicer_amd import build as jbuild
    from oracle import oracle as orc
    jbuild.build()
    orc.build()
    return True
