"""BatchNorm running statistics (the buffers the reference's checkpoints carry, utils/experiment.py:85-90): the engine's export
of per-pass batch statistics + the closed-form fold against torch.nn.functional.batch_norm updating shared buffers pass by pass
in the reference's call order (learn2learn clones share buffers; every learner(x) of maml_vision.py:102-124 updates them)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from exploring_meta_amd.core_functions.vision_models import (MiniImagenetCNN, OmniglotCNN, RunningStatsFold, apply_running_stats,
                                                              running_stats_contribution)
from exploring_meta_amd.engine import MetaEngine, ModelSpec
from exploring_meta_amd.utils import synthetic
from oracle import vision_ref as R

pytestmark = pytest.mark.gpu


def _unflatten(flat, shapes):
    out, off = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def _forward_tracking(x, p, base, running):
    """conv_base of the oracle with torch's own running-statistics update (the side effect under test)."""
    stride = int(2 * base['max_pool_factor'])
    for i in range(base['layers']):
        x = F.conv2d(x, p[f'base.{i}.conv.weight'], p[f'base.{i}.conv.bias'], stride=1 if base['max_pool'] else stride, padding=1)
        rm, rv = running[i]
        x = F.relu(F.batch_norm(x, rm, rv, p[f'base.{i}.normalize.weight'], p[f'base.{i}.normalize.bias'], training=True, momentum=0.1, eps=1e-5))
        if base['max_pool']:
            x = F.max_pool2d(x, stride, stride)
    return x


@pytest.mark.parametrize('dataset,ways,shots,K,T', [('min', 5, 1, 2, 3), ('omni', 5, 1, 3, 4)])
def test_running_stats_match_torch_batchnorm_pass_by_pass(dataset, ways, shots, K, T):
    spec = R.mini_imagenet_spec(ways) if dataset == 'min' else R.omniglot_spec(ways)
    model = (MiniImagenetCNN(ways) if dataset == 'min' else OmniglotCNN(ways)).cuda()
    mspec = model.spec()
    shapes = R.param_shapes(spec)
    theta = model.flat_parameters().detach().contiguous()
    data, labels = synthetic.make_meta_batch(dataset, list(range(T)), ways, shots)
    eng = MetaEngine(mspec)
    trace = eng.set_trace(T, K)
    fold = RunningStatsFold(eng, model.base, mspec, T, 0, T, K + 1, ways * shots, phases=1)
    eng.meta_batch(theta, torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), shots, K, 0.4)
    torch.cuda.synchronize()
    fold.collect(0)
    eng.set_trace(0)
    # the reference's side effect: shared buffers, updated by every forward pass in call order (fp64, at the engine's own theta_k)
    running = [[torch.zeros(spec['base']['hidden'], dtype=torch.float64), torch.ones(spec['base']['hidden'], dtype=torch.float64)]
               for _ in range(spec['base']['layers'])]
    for t in range(T):
        xs, _, xq, _ = R.prepare_batch(torch.from_numpy(data[t]).double(), torch.from_numpy(labels[t]), shots, ways)
        for k in range(K + 1):
            p = {n: v.double() for n, v in _unflatten(trace['theta'][k, t].cpu(), shapes).items()}
            _forward_tracking(xs if k < K else xq, p, spec['base'], running)
    fold.apply()
    assert eng._bn_export is None
    for i, blk in enumerate(model.base):
        assert int(blk.normalize.num_batches_tracked) == T * (K + 1)
        for got, want in ((blk.normalize.running_mean, running[i][0]), (blk.normalize.running_var, running[i][1])):
            err = float((got.double().cpu() - want).abs().max() / want.abs().max())
            assert err < 2e-5, (i, err)
    assert model.state_dict()['base.0.normalize.running_var'].abs().sum() > 0
