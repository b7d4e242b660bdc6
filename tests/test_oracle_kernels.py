"""oracle/kernels_ref.py (the engine's decomposition: explicit backward + forward-over-reverse HVP adjoint recursion)
against oracle/vision_ref.py (autograd restatement of the reference), fp64."""
import numpy as np
import pytest
import torch

from oracle import kernels_ref as KR
from oracle import vision_ref as R
from helpers import model_params, task_tensors, rel_err


def _run(spec, dataset, ways, shots, K, lr, fo, tasks, seed=11):
    theta = model_params(spec, seed)
    datas, labels = task_tensors(dataset, tasks, ways, shots)
    losses, accs, grad, logits = R.maml_meta_batch(theta, spec, datas, labels, K, shots, ways, lr, fo)
    desc = KR.net_desc(spec)
    th_e = KR.to_engine_params(theta, desc)
    tot = None
    for t, (d, l) in enumerate(zip(datas, labels)):
        si, qi = R.prepare_batch_indices(d.shape[0], shots, ways)
        x = KR.nchw_to_nhwc(d.view(-1, *spec['in_shape']))
        loss, acc, lam, lg = KR.maml_task(th_e, x[si], l[si], x[qi], l[qi], desc, K, lr, fo)
        assert loss.item() == pytest.approx(losses[t].item(), rel=1e-9, abs=1e-12)
        assert acc.item() == accs[t].item()
        assert np.allclose(lg.numpy(), logits[t].numpy(), rtol=1e-8, atol=1e-9)
        tot = lam if tot is None else [a + b for a, b in zip(tot, lam)]
    g = KR.from_engine_grads(tot, desc, list(theta.keys()))
    for k in theta:
        if k.endswith('conv.bias'):
            assert grad[k].abs().max().item() < 1e-9          # inert under batch-stat BN
            continue
        assert rel_err(g[k].numpy(), grad[k].numpy()) < 1e-7, k
    return g


def test_min_first_order_one_step():
    _run(R.mini_imagenet_spec(5), 'min', 5, 1, 1, 0.5, True, [0])


def test_min_second_order_one_step():
    _run(R.mini_imagenet_spec(5), 'min', 5, 1, 1, 0.5, False, [0, 1])


def test_min_second_order_two_steps():
    _run(R.mini_imagenet_spec(5), 'min', 5, 1, 2, 0.1, False, [2])


def test_omni_second_order_two_steps():
    _run(R.omniglot_spec(5), 'omni', 5, 1, 2, 0.4, False, [0])


def test_omni_first_order():
    _run(R.omniglot_spec(5), 'omni', 5, 1, 1, 0.5, True, [0, 1])


def test_param_layout_roundtrip():
    spec = R.mini_imagenet_spec(5)
    theta = model_params(spec, 3)
    desc = KR.net_desc(spec)
    back = KR.from_engine_grads(KR.to_engine_params(theta, desc), desc, list(theta.keys()))
    for k in theta:
        assert torch.equal(back[k], theta[k])
