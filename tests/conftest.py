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


@pytest.fixture(params=['split_f16', 'split_bf16', 'fp32_pipe'])
def conv_form(request):
    """Operand form of the 32- / 64-channel stride-1 convolutions and weight gradients (mi_conv_set_split_bf16) for the duration of one test:
    the exact three-plane bf16 form (the default), the fp32 matrix pipe, and the opt-in two-plane fp16 form -- every bar holds for each form on
    its own, so a regression in one is not absorbed by another's envelope."""
    from exploring_meta_amd import _lib
    lb = _lib.load()
    was = lb.mi_conv_set_split_bf16({'split_f16': 2, 'split_bf16': 1, 'fp32_pipe': 0}[request.param])
    yield request.param
    lb.mi_conv_set_split_bf16(was)
