"""The reference's call surface on the GPU: the loop body of vision/maml_vision.py:102-114 written against this package
(MAML(model).clone() -> fast_adapt -> eval_loss.backward()) must leave in ``model.parameters()``'s ``.grad`` what the
reference leaves there (golden fixtures), and ``evaluate`` / ``model(x)`` / ``prepare_batch`` must agree with the oracle."""
import numpy as np
import pytest
import torch

from exploring_meta_amd import core_functions as cf
from exploring_meta_amd.utils import synthetic
from oracle import vision_ref as R
from helpers import model_params, task_tensors
from gpu_utils import rel_err, report

pytestmark = pytest.mark.gpu


def _load(model, theta):
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(theta[k].float())
    return model.cuda()


def test_reference_loop_body_cfg4(golden_fa):
    tag, ways, shots, K, lr, tasks = 'cfg4_min_5w1s_K1_so', 5, 1, 1, 0.5, [0, 1, 2]
    spec = R.mini_imagenet_spec(ways)
    model = _load(cf.MiniImagenetCNN(ways), model_params(spec, 11))
    maml = cf.MAML(model, lr=lr, first_order=False)
    loss_fn = torch.nn.CrossEntropyLoss(reduction='mean')
    device = torch.device('cuda')
    meta_train_loss, meta_train_acc = 0.0, 0.0
    for t in tasks:                                           # reference: for task in range(meta_batch_size)
        learner = maml.clone()
        d, l = synthetic.make_task('min', t, ways, shots)
        batch = (torch.from_numpy(d), torch.from_numpy(l))
        eval_loss, eval_acc = cf.fast_adapt(batch, learner, loss_fn, K, shots, ways, device)
        eval_loss.backward()
        meta_train_loss += eval_loss.item()
        meta_train_acc += eval_acc.item()
    grad = torch.cat([p.grad.reshape(-1) for p in maml.parameters()]).cpu().numpy()
    gold = golden_fa[f'g3_{tag}_f64_grad']
    e = rel_err(grad, gold)
    report('surface_loop_cfg4', grad_rel=e, loss_sum=meta_train_loss, loss_sum_ref=float(golden_fa[f'g3_{tag}_f64_loss'].sum()))
    assert e < 3e-2                                           # reference fp32 itself: 1.0e-2 on these tasks
    assert meta_train_loss == pytest.approx(golden_fa[f'g3_{tag}_f64_loss'].sum(), rel=1e-4)
    assert meta_train_acc == pytest.approx(golden_fa[f'g3_{tag}_f64_acc'].sum())


def test_reference_loop_body_omniglot_first_order(golden_fa):
    tag, ways, shots, K, lr, tasks = 'cfg1_omni_5w1s_K1_fo', 5, 1, 1, 0.5, [0, 1]
    model = _load(cf.OmniglotCNN(ways), model_params(R.omniglot_spec(ways), 11))
    maml = cf.MAML(model, lr=lr, first_order=True)
    loss_fn = torch.nn.CrossEntropyLoss()
    for t in tasks:
        d, l = synthetic.make_task('omni', t, ways, shots)
        eval_loss, _ = cf.fast_adapt((torch.from_numpy(d), torch.from_numpy(l)), maml.clone(), loss_fn, K, shots, ways,
                                     torch.device('cuda'))
        eval_loss.backward()
    grad = torch.cat([p.grad.reshape(-1) for p in maml.parameters()]).cpu().numpy()
    assert rel_err(grad, golden_fa[f'g3_{tag}_f64_grad']) < 1e-4


def test_deferred_outer_backward_gives_the_same_gradient(monkeypatch):
    """core_functions.vision.DEFERRED_OUTER_BACKWARD (for loops that adapt with gradients enabled but do not always call backward,
    like the reference's validation half): evaluation-only forward, the fused call with the meta-gradient re-run in backward --
    same loss, same accuracy, bit-identical gradient; a loss that is never back-propagated costs no outer backward."""
    from exploring_meta_amd.core_functions import vision as V
    ways, shots, K = 5, 1, 2
    out = {}
    for mode in (False, True):
        monkeypatch.setattr(V, 'DEFERRED_OUTER_BACKWARD', mode)
        model = _load(cf.MiniImagenetCNN(ways), model_params(R.mini_imagenet_spec(ways), 11))
        maml = cf.MAML(model, lr=0.3, first_order=False)
        d, l = synthetic.make_meta_batch('min', [0, 1, 2], ways, shots)
        total, losses, accs = cf.meta_batch_adapt(maml.clone(), torch.from_numpy(d).cuda(), torch.from_numpy(l).cuda(), K, shots, ways)
        total.backward()
        out[mode] = (losses.cpu(), accs.cpu(), torch.cat([p.grad.reshape(-1) for p in maml.parameters()]).cpu())
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
    assert torch.equal(out[False][2], out[True][2])


def test_model_forward_matches_reference(golden_small):
    model = _load(cf.MiniImagenetCNN(5), model_params(R.mini_imagenet_spec(5), 7))
    data, _ = synthetic.make_task('min', 3, 5, 5)
    y = model(torch.from_numpy(data).cuda()).detach().cpu().numpy()
    ref = golden_small['g2_min32_f64_out']
    assert np.max(np.abs(y - ref)) < 1e-4 * max(1.0, np.abs(ref).max())
    omni = _load(cf.OmniglotCNN(5), model_params(R.omniglot_spec(5), 7))
    data, _ = synthetic.make_task('omni', 3, 5, 1)
    y = omni(torch.from_numpy(data).cuda()).detach().cpu().numpy()
    ref = golden_small['g2_omni64_f64_out']
    assert np.max(np.abs(y - ref)) < 1e-4 * max(1.0, np.abs(ref).max())


class _Sampler:
    def __init__(self, dataset, ways, shots):
        self.dataset, self.ways, self.shots, self.i = dataset, ways, shots, 0

    def sample(self):
        d, l = synthetic.make_task(self.dataset, self.i, self.ways, self.shots)
        self.i += 1
        return torch.from_numpy(d), torch.from_numpy(l)


def test_evaluate_matches_oracle():
    ways, shots, K, lr, T = 5, 1, 1, 0.5, 6
    spec = R.mini_imagenet_spec(ways)
    theta = model_params(spec, 11)
    model = _load(cf.MiniImagenetCNN(ways), theta)
    params = dict(meta_batch_size=T, adapt_steps=K, shots=shots, ways=ways)
    acc = cf.evaluate(params, _Sampler('min', ways, shots), cf.MAML(model, lr=lr), torch.nn.CrossEntropyLoss(), torch.device('cuda'))
    datas, labels = task_tensors('min', list(range(T)), ways, shots)
    _, accs, _, _ = R.maml_meta_batch(theta, spec, datas, labels, K, shots, ways, lr, False, backward=False)
    assert acc == pytest.approx(accs.mean().item(), abs=1e-6)
    assert all(p.grad is None for p in model.parameters())


def test_prepare_batch_cuda_path(golden_small):
    ways, shots = 5, 5
    d, l = synthetic.make_task('min', 0, ways, shots)
    ad, al, ed, el = cf.prepare_batch((torch.from_numpy(d), torch.from_numpy(l)), shots, ways, torch.device('cuda'))
    si, qi = golden_small['g1_5w5s_support_rows'], golden_small['g1_5w5s_query_rows']
    assert torch.equal(ad.cpu(), torch.from_numpy(d[si])) and torch.equal(ed.cpu(), torch.from_numpy(d[qi]))
    assert torch.equal(al.cpu(), torch.from_numpy(l[si])) and torch.equal(el.cpu(), torch.from_numpy(l[qi]))
