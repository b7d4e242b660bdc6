import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
