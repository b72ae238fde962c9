import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_objects(request):
    """After every GPU test: drop the dead modules' captured hipGraphs, streams and buffers now, not whenever the
    garbage collector gets to them - a long session otherwise accumulates dozens of instantiated graphs (observed on
    ROCm 7.2: with test_step_gpu.py run BEFORE test_fullsize_gpu.py, the C5 test's graph replay segfaulted in
    hipGraphLaunch; with the dead captures released after each test every order passes)."""
    yield
    if "gpu" in request.keywords and os.environ.get("TACORL_TEST_NO_RELEASE") != "1":  # (=1: reproduce the crash this fixture avoids)
        import gc

        import torch

        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
