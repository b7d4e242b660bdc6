"""``fast_adapt`` / ``accuracy`` / ``evaluate`` with the reference's signatures (core_functions/vision.py:6-42), backed by
the batched HIP engine, plus the batched entry the engine is designed for (``meta_batch_adapt``)."""
import os

import torch

from ..engine import flatten_parameters
from .maml import MAML


def _check_loss(loss):
    if not isinstance(loss, torch.nn.CrossEntropyLoss) or loss.reduction != 'mean' or loss.weight is not None \
            or getattr(loss, 'label_smoothing', 0.0) != 0.0:
        raise ValueError("the HIP engine fuses torch.nn.CrossEntropyLoss(reduction='mean'), the loss every reference "
                         'vision script uses (maml_vision.py:86, anil_vision.py:99)')


# The meta-gradient is normally produced WITH the loss (one fused call).  A caller that adapts under enabled gradients but never
# calls backward -- the reference's validation half, maml_vision.py:117-124, which is not wrapped in no_grad -- then pays for an
# outer backward it discards (about 2/3 of a second-order call).  With this switch on (or MI_MAML_DEFERRED_BACKWARD=1) the forward
# runs the evaluation-only call and `backward` re-runs the fused call with the gradient: cheaper when fewer than about two thirds
# of the calls are followed by backward, dearer otherwise (the drivers here wrap validation in no_grad instead and leave it off).
DEFERRED_OUTER_BACKWARD = os.environ.get('MI_MAML_DEFERRED_BACKWARD', '0') == '1'


class _FusedFastAdapt(torch.autograd.Function):
    """T tasks through mi_meta_batch_maml.  The meta-gradient is produced together with the loss (the outer backward is
    part of the fused call); ``backward`` hands it to autograd so ``eval_loss.backward()`` accumulates into ``.grad``."""

    @staticmethod
    def forward(ctx, engine, data, labels, shots, steps, lr, first_order, need_grad, grad_tasks, *params):
        theta = torch.cat([p.detach().reshape(-1) for p in params]).float().contiguous()
        ctx.deferred = None
        gt = None if grad_tasks is None else int(grad_tasks)
        if need_grad and DEFERRED_OUTER_BACKWARD:
            ctx.deferred = (engine, theta, data, labels, shots, steps, lr, first_order, gt)
            need_grad = False
        loss, acc, grad, _ = engine.meta_batch(theta, data, labels, shots, steps, lr, first_order=first_order,
                                               with_grad=need_grad, grad_tasks=gt if need_grad else None)
        ctx.shapes = [p.shape for p in params]
        ctx.per_task = None
        ctx.save_for_backward(grad if grad is not None else torch.empty(0, device=data.device))
        # Only the SUM carries the meta-gradient (the engine reduces over tasks inside the fused call): the per-task losses are
        # values, so `losses.mean().backward()` fails loudly instead of stepping on zeros.
        ctx.mark_non_differentiable(loss, acc)
        return (loss.sum() if gt is None else loss[:gt].sum()), loss, acc

    @staticmethod
    def backward(ctx, gsum, gloss, gacc):
        (grad,) = ctx.saved_tensors
        if ctx.deferred is not None:
            engine, theta, data, labels, shots, steps, lr, first_order, gt = ctx.deferred
            grad = engine.meta_batch(theta, data, labels, shots, steps, lr, first_order=first_order, with_grad=True, grad_tasks=gt)[2]
        if grad.numel() == 0:
            raise RuntimeError('fast_adapt was run without gradients (torch.no_grad or no parameter requires grad)')
        outs, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for d in shp:
                n *= d
            outs.append((grad[off:off + n] * gsum).reshape(shp))
            off += n
        return (None,) * 9 + tuple(outs)


def meta_batch_adapt(learner, data, labels, adaptation_steps, shots, ways, first_order=None, grad_tasks=None):
    """Batched entry: data [T, 2*shots*ways, C, H, W], labels [T, 2*shots*ways] on the GPU.
    Returns (loss_sum, loss[T], acc[T]); ``loss_sum.backward()`` accumulates the SUM over tasks of d valid_loss/d theta
    into the base parameters' ``.grad`` -- what T iterations of the reference loop body leave there.  ``loss[T]`` and ``acc[T]``
    are plain values (not differentiable): weight tasks by scaling ``loss_sum``, or call once per group of tasks.
    ``grad_tasks = G``: the first G tasks are the iteration's train tasks (``loss_sum`` and the gradient cover them), the others its
    validation tasks (reference maml_vision.py:117-124), adapted and scored in the same launches without a backward half."""
    model = learner.module if isinstance(learner, MAML) else learner
    fo = learner.first_order if first_order is None and isinstance(learner, MAML) else bool(first_order)
    lr = learner.lr
    params = list(model.parameters())
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    spec = model.spec()
    if spec.ways != ways:
        raise ValueError(f'model has {spec.ways} outputs but ways={ways}')
    data = data.reshape(data.shape[0], data.shape[1], spec.in_channels, spec.in_h, spec.in_w).float().contiguous()
    return _FusedFastAdapt.apply(model.engine(), data, labels.contiguous(), shots, adaptation_steps, lr, fo, need, grad_tasks, *params)


def fast_adapt(batch, learner, loss, adaptation_steps, shots, ways, device, features=None):
    """Same signature/returns as the reference (vision.py:6-18): (valid_loss, valid_accuracy) as 0-dim tensors;
    ``valid_loss.backward()`` accumulates the (second-order unless learner.first_order) meta-gradient into the base
    module's parameters."""
    _check_loss(loss)
    data, labels = batch
    data, labels = data.to(device), labels.to(device)
    if features is not None:
        from .anil import fast_adapt_anil
        return fast_adapt_anil(data, labels, learner, features, adaptation_steps, shots, ways)
    total, losses, accs = meta_batch_adapt(learner, data.unsqueeze(0), labels.unsqueeze(0), adaptation_steps, shots, ways)
    return total, accs[0]


def accuracy(predictions, targets):
    """reference vision.py:21-23"""
    predictions = predictions.argmax(dim=1).view(targets.shape)
    return (predictions == targets).sum().float() / targets.size(0)


def evaluate(params, test_tasks, model, loss, device, features=None):
    """reference vision.py:26-42: mean query accuracy over params['meta_batch_size'] test tasks (no backward).  The tasks are
    sampled in the reference's order and then adapted in ONE batched engine call."""
    _check_loss(loss)
    batches = [test_tasks.sample() for _ in range(params['meta_batch_size'])]
    with torch.no_grad():
        if features is None:
            data = torch.stack([b[0] for b in batches]).to(device)
            labels = torch.stack([b[1] for b in batches]).to(device)
            _, _, accs = meta_batch_adapt(model.clone(), data, labels, params['adapt_steps'], params['shots'], params['ways'])
            meta_test_accuracy = accs.mean().item()
        else:
            meta_test_accuracy = 0.0
            for b in batches:
                _, acc = fast_adapt(b, model.clone(), loss, params['adapt_steps'], params['shots'], params['ways'], device,
                                    features=features)
                meta_test_accuracy += acc.item()
            meta_test_accuracy /= params['meta_batch_size']
    print('Meta Test Accuracy', meta_test_accuracy)
    return meta_test_accuracy
