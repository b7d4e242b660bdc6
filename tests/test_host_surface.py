"""Host-side mirror of the reference interface (no GPU needed): state_dict layout, prepare_batch rows, accuracy, the MAML
wrapper's clone semantics, the C-ABI symbol table, and that the product refuses to compute without the GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from exploring_meta_amd import _lib
from exploring_meta_amd import core_functions as cf
from exploring_meta_amd.engine import ModelSpec
from exploring_meta_amd.utils import synthetic

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name,ctor', [('min32', lambda: cf.MiniImagenetCNN(5)), ('omni64', lambda: cf.OmniglotCNN(5)),
                                       ('base_min64', lambda: cf.ConvBase(output_size=64, channels=3, max_pool=True))])
def test_state_dict_layout_matches_reference(golden_small, name, ctor):
    m = ctor()
    sd = m.state_dict()
    assert list(sd.keys()) == list(golden_small[f'g6_{name}_keys'])
    assert [','.join(str(d) for d in v.shape) for v in sd.values()] == list(golden_small[f'g6_{name}_shapes'])
    assert [k for k, _ in m.named_parameters()] == list(golden_small[f'g6_{name}_param_names'])


def test_model_spec_param_shapes_match_modules():
    for m in (cf.MiniImagenetCNN(5), cf.OmniglotCNN(20)):
        assert [(k, tuple(p.shape)) for k, p in m.named_parameters()] == m.spec().param_shapes()


def test_initialisers_follow_reference():
    torch.manual_seed(0)
    m = cf.MiniImagenetCNN(5)
    for blk in m.base:
        assert float(blk.normalize.weight.min()) >= 0.0 and float(blk.normalize.weight.max()) <= 1.0   # uniform_(0,1)
        assert torch.count_nonzero(blk.conv.bias) == 0 and torch.count_nonzero(blk.normalize.bias) == 0
        fan = blk.conv.weight.shape[1] * 9 + blk.conv.weight.shape[0] * 9
        assert float(blk.conv.weight.abs().max()) <= (6.0 / fan) ** 0.5 + 1e-6                          # xavier_uniform_
    assert torch.count_nonzero(m.linear.bias) == 0
    o = cf.OmniglotCNN(5)
    assert 0.5 < float(o.linear.weight.std()) < 1.5                                                    # normal_()


@pytest.mark.parametrize('ways,shots', [(5, 1), (5, 5), (20, 1), (20, 5)])
def test_prepare_batch_host(golden_small, ways, shots):
    n = 2 * shots * ways
    data = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 1, 2, 2).contiguous()
    labels = torch.from_numpy(synthetic.task_labels(ways, shots))
    ad, al, ed, el = cf.prepare_batch((data, labels), shots, ways, torch.device('cpu'))
    assert np.array_equal(ad[:, 0, 0, 0].numpy().astype(np.int64), golden_small[f'g1_{ways}w{shots}s_support_rows'])
    assert np.array_equal(ed[:, 0, 0, 0].numpy().astype(np.int64), golden_small[f'g1_{ways}w{shots}s_query_rows'])
    assert np.array_equal(al.numpy(), golden_small[f'g1_{ways}w{shots}s_support_labels'])
    assert np.array_equal(el.numpy(), golden_small[f'g1_{ways}w{shots}s_query_labels'])
    with pytest.raises(ValueError):
        cf.prepare_batch((data[:-1], labels[:-1]), shots, ways, torch.device('cpu'))


def test_accuracy_ties(golden_small):
    acc = cf.accuracy(torch.from_numpy(golden_small['g4_preds']), torch.from_numpy(golden_small['g4_targets']))
    assert acc.item() == golden_small['g4_acc'][0]


def test_maml_wrapper_surface():
    model = cf.OmniglotCNN(5)
    maml = cf.MAML(model, lr=0.5, first_order=False)
    assert [id(p) for p in maml.parameters()] == [id(p) for p in model.parameters()]
    learner = maml.clone()
    assert isinstance(learner, cf.MAML) and learner.module is model and learner.lr == 0.5 and learner.first_order is False
    assert maml.clone(first_order=True).first_order is True
    from exploring_meta_amd.algorithms import MAML as by_l2l_path           # `from learn2learn.algorithms import MAML` with the package swapped
    from exploring_meta_amd.algorithms.maml import MAML as by_base_path     # reference core_functions/maml.py:8
    assert by_l2l_path is cf.MAML and by_base_path is cf.MAML
    assert learner.hidden_size == 64                       # attribute forwarding to the wrapped module
    assert set(maml.state_dict().keys()) == {'module.' + k for k in model.state_dict().keys()}
    fw = learner.fast_weights()                            # flat, parameters() order, graph-connected to the base parameters
    assert fw.shape == (sum(p.numel() for p in model.parameters()),) and fw.requires_grad
    assert learner.clone().fast_weights() is fw            # a clone starts from the learner's current weights
    with pytest.raises(RuntimeError):                      # a loss that does not come from learner(x)
        learner.adapt(torch.tensor(0.0))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                  # step-wise forward is GPU-only as well
            learner(torch.zeros(5, 1, 28, 28))


def test_loss_must_be_mean_cross_entropy():
    model = cf.OmniglotCNN(5)
    batch = (torch.zeros(10, 1, 28, 28), torch.from_numpy(synthetic.task_labels(5, 1)))
    with pytest.raises(ValueError):
        cf.fast_adapt(batch, cf.MAML(model, 0.5).clone(), torch.nn.MSELoss(), 1, 1, 5, torch.device('cpu'))


def test_no_cpu_fallback():
    """The product path must fail loudly without the GPU -- never fall back to a CPU implementation."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    model = cf.OmniglotCNN(5)
    batch = (torch.zeros(10, 1, 28, 28), torch.from_numpy(synthetic.task_labels(5, 1)))
    with pytest.raises(RuntimeError):
        cf.fast_adapt(batch, cf.MAML(model, 0.5).clone(), torch.nn.CrossEntropyLoss(), 1, 1, 5, torch.device('cpu'))
    with pytest.raises(RuntimeError):
        model(torch.zeros(5, 1, 28, 28))


def test_library_exports_every_declared_symbol():
    """include/mi_maml.h <-> libmi_maml.so <-> the ctypes table."""
    header = open(os.path.join(REPO, 'include', 'mi_maml.h')).read()
    declared = set(re.findall(r'\b(mi_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/mi_maml.h but not exported'
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'exploring_meta_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f'{f} imports the oracle'


def test_roofline_names_the_kernel_a_conv_launch_takes():
    """utils/roofline.py mirrors launch_conv3x3's rule for the split-bf16 form's two kernels (csrc/conv_mfma.hip: conv_b16_for): the 16x16x32
    kernel from 6 tiles of 30 pixels per wave (rounded up; 2048 resident waves) on in mode 1, always / never in modes 2 / 0 -- bench.py names
    the dominant kernel with it."""
    from exploring_meta_amd.engine import ModelSpec
    from exploring_meta_amd.utils import roofline as RF
    # cfg2, 32 tasks x 25 images: block 2 (42 x 42) 23 tiles per wave, block 3 (21 x 21) 5.7, block 4 (10 x 10) 1.3
    assert RF.conv_kernel_is_b16(25 * 42 * 42, 32, 32, 1) and RF.conv_kernel_is_b16(25 * 21 * 21, 32, 32, 1)
    assert not RF.conv_kernel_is_b16(25 * 10 * 10, 32, 32, 1)
    # few tasks per call stay on the 32x32x16 kernel (4 tasks: block 2 at 2.9 tiles per wave; 8 tasks: 5.7 -> 6: the new kernel); cfg4
    # (5 images per task: block 2 at 4.6 -> 5) stays too; modes 0 / 2 do not look at the size
    assert not RF.conv_kernel_is_b16(25 * 42 * 42, 4, 32, 1) and RF.conv_kernel_is_b16(25 * 42 * 42, 8, 32, 1)
    assert not RF.conv_kernel_is_b16(5 * 42 * 42, 32, 32, 1)
    assert RF.conv_kernel_is_b16(5 * 10 * 10, 1, 32, 2) and not RF.conv_kernel_is_b16(25 * 42 * 42, 32, 32, 0)
    # 64 filters: two channel tiles per pixel tile
    assert RF.conv_kernel_is_b16(50 * 21 * 21, 32, 64, 1)
    spec = ModelSpec.mini_imagenet(5)
    assert RF.kernel_name(spec, 'tangent_conv_fwd', 1, b16=True).startswith('conv3x3_s1_b16_kernel<32,2,EPI_TSTATS')
    assert RF.kernel_name(spec, 'tangent_conv_fwd', 1).startswith('conv3x3_s1_mfma_kernel<32,2,EPI_TSTATS')
    assert RF.kernel_name(spec, 'wgrad', 1, b16=True).startswith('wgrad3x3')          # only the forward / dgrad family has the second kernel


def test_time_feature_without_division_is_the_quotient():
    """csrc/gae.hip::gae_time: t / 100.0 as q = RN(t * 0.01), r = fma(-q, 100, t), fma(r, 0.01, q) -- the LinearValue time features
    (reference rl.py:95-110 via cherry's LinearValue: t = row / 100) must be bit for bit those of the division.  Exact rational arithmetic
    for every row index a replay can hold (mi_gae_max_rows is below 6000) and well beyond."""
    from fractions import Fraction
    fma = lambda a, b, c: float(Fraction(a) * Fraction(b) + Fraction(c))     # one rounding, like the hardware FMA
    for t in list(range(0, 20000)) + [2 ** k + d for k in range(15, 24) for d in (-1, 0, 1, 37)]:
        td = float(t)
        q = td * 0.01
        assert fma(fma(-q, 100.0, td), 0.01, q) == td / 100.0, t
