import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gi-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("GIGS_TEST_NOGC") == "1":  # diagnostic: no cyclic-GC passes at arbitrary points of the session
        import gc
        gc.disable()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as _orc
    _orc.build()
    _orc.set_threads(min(8, _orc.max_threads()))
    return _orc
