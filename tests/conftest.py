import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle legs are many small fp64 products: on the GPU box's 256-logical-core host torch's default of 128 intra-op
    # threads makes them 4-5x slower than 16 (measured: the cfg5 full-size test 61 s -> 12 s), and the suite has a wall-clock budget.
    import torch
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, cores)))


@pytest.fixture(scope='session')
def golden_small():
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'golden_small.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_fa():
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'golden_fast_adapt.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_refinit():
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'golden_refinit.npz'), allow_pickle=False)
