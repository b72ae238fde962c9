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


# (Rounds 3-5 had an autouse fixture here that released dead modules' hipGraphs, streams and buffers after every GPU test:
# in round 3 `hipGraphLaunch` had segfaulted with tests/test_step_gpu.py run before tests/test_fullsize_gpu.py.  Round 5
# could not reproduce it (48 live graphs replay fine, scratch/graph_stress.py); round 6 ran the whole GPU suite three times
# without the fixture - TACORL_TEST_NO_RELEASE=1, scratch/r6_norelease.sh: 239 passed each time, in 10 minutes instead of
# 14.5 - and removed it.  What protects a capture from being replayed against freed memory is the allocation-epoch stamp
# (ops.note_alloc / GraphMixin._run_segments), not a garbage-collection schedule.)
