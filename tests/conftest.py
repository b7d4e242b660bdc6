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
def golden_rl():
    """Records of the REFERENCE's rl.py:95-110,346-473 executed on seeded replays (tests/golden/make_golden_rl.py)."""
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'golden_rl.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_refinit():
    import numpy as np
    return np.load(os.path.join(REPO, 'tests', 'golden', 'golden_refinit.npz'), allow_pickle=False)


# Operand form / kernel of the 32- / 64-channel stride-1 convolutions and weight gradients: name -> (mi_conv_set_split_bf16, mi_conv_set_b16).
# 'split_bf16' is the engine's default (exact three-plane bf16 operands; the 16x16x32 kernel of csrc/conv_b16.h for launches of >= 6 tiles
# per wave, the 32x32x16 kernel of rounds 3-4 below that -- what bench.py times), 'split_bf16_16x16' / 'split_bf16_32x32' the same operand
# form with EVERY launch on one of the two kernels (small test shapes reach the 16x16x32 kernel only this way), 'fp32_pipe' the fp32 matrix
# pipe, 'split_f16' the opt-in two-plane fp16 form.
CONV_FORMS = {'split_f16': (2, -1), 'split_bf16': (1, 1), 'split_bf16_16x16': (1, 2), 'split_bf16_32x32': (1, 0), 'fp32_pipe': (0, -1)}


def apply_conv_form(lb, name):
    """Select a form; returns a callable that restores the previous selection."""
    form, b16 = CONV_FORMS[name]
    was = lb.mi_conv_set_split_bf16(form)
    was16 = lb.mi_conv_set_b16(b16)

    def restore():
        lb.mi_conv_set_split_bf16(was)
        lb.mi_conv_set_b16(was16)
    return restore


@pytest.fixture(params=['split_f16', 'split_bf16', 'split_bf16_16x16', 'fp32_pipe'])
def conv_form(request):
    """Operand form of the 32- / 64-channel stride-1 convolutions and weight gradients for the duration of one test: every bar holds for
    each form on its own, so a regression in one is not absorbed by another's envelope."""
    from exploring_meta_amd import _lib
    restore = apply_conv_form(_lib.load(), request.param)
    yield request.param
    restore()


@pytest.fixture(params=['split_bf16', 'fp32_pipe'])
def conv_form_full(request):
    """The forms the long full-size tests run (cfg2 teacher-forced, cfg3: their oracle legs take a minute of CPU per form): the shipped operand
    form and the fp32 pipe.  The opt-in fp16 form keeps its full-size coverage in the one-step configuration (conv_form_full3) and all of
    its kernel-level coverage (conv_form); the suite has to fit the driver's step budget (round 5: 533 s of 900)."""
    from exploring_meta_amd import _lib
    restore = apply_conv_form(_lib.load(), request.param)
    yield request.param
    restore()


@pytest.fixture(params=['split_f16', 'split_bf16', 'fp32_pipe'])
def conv_form_full3(request):
    """All three operand forms, for the short full-size tests."""
    from exploring_meta_amd import _lib
    restore = apply_conv_form(_lib.load(), request.param)
    yield request.param
    restore()
