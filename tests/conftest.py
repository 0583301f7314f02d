import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gi-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as _orc
    _orc.build()
    _orc.set_threads(min(8, _orc.max_threads()))
    return _orc


@pytest.fixture(autouse=True)
def _gpu_teardown(request):
    """GPU tests: objects that own hipGraphs (pipeline.WholeStepGraph <-> its Stage2Step: a reference cycle) die HERE, at
    the test boundary with the device idle, not at whatever later moment a cyclic-GC pass happens to run."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
