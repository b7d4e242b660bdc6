#!/usr/bin/env python3
"""Counterpart of the reference's ``vision/maml_vision.py`` on the batched HIP engine.

Same experiment (seeds, model, ``MAML(model, lr, first_order)``, Adam(outer_lr), CrossEntropy, per-iteration train +
validation meta-batches, gradient / meta_batch_size, meta-test with ``evaluate``), same flags plus ``--first_order`` (the
reference hard-codes second order, maml_vision.py:84) -- but the ``for task in range(meta_batch_size)`` loop
(maml_vision.py:102-124) is ONE ``meta_batch_adapt`` call for both halves (train tasks with, validation tasks without the backward
half: ``grad_tasks``), and with ``torchrun`` the meta-batch is sharded over
ranks with one RCCL all-reduce.  Datasets are not available offline: tasks come from the seeded synthetic generator
(``utils/synthetic.py``) with the reference's batch layout.

    python -m exploring_meta_amd.vision.maml_vision --dataset min --shots 5 --adapt_steps 5 --num_iterations 10
"""
import argparse
import os
import random

import numpy as np
import torch

from ..core_functions import MAML, MiniImagenetCNN, OmniglotCNN, evaluate, meta_batch_adapt
from ..core_functions.vision_models import RunningStatsFold
from ..sharding import init_process_group, reduce_meta_batch, shard_range
from ..utils import synthetic

params = {
    "ways": 5, "shots": 1, "outer_lr": 0.003, "inner_lr": 0.5, "adapt_steps": 1, "meta_batch_size": 32,
    "num_iterations": 10000, "save_every": 1000, "seed": 42,
}


class SyntheticTasks:
    """Stand-in for an l2l TaskDataset: ``sample()`` returns (data [2*S*W,C,H,W], labels [2*S*W])."""

    def __init__(self, dataset, ways, shots, first_id):
        self.dataset, self.ways, self.shots, self.next = dataset, ways, shots, first_id

    def sample(self):
        d, l = synthetic.make_task(self.dataset, self.next, self.ways, self.shots)
        self.next += 1
        return torch.from_numpy(d), torch.from_numpy(l)

    def sample_batch(self, ids):
        d, l = synthetic.make_meta_batch(self.dataset, ids, self.ways, self.shots)
        return torch.from_numpy(d), torch.from_numpy(l)


def run(dataset, p, first_order=False, log=print):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        init_process_group(local)
    random.seed(p['seed']); np.random.seed(p['seed']); torch.manual_seed(p['seed']); torch.cuda.manual_seed(p['seed'])
    device = torch.device('cuda', local)
    model = (OmniglotCNN(p['ways']) if dataset == 'omni' else MiniImagenetCNN(p['ways'])).to(device)
    maml = MAML(model, lr=p['inner_lr'], first_order=first_order)
    opt = torch.optim.Adam(maml.parameters(), p['outer_lr'])
    loss = torch.nn.CrossEntropyLoss(reduction='mean')
    T = p['meta_batch_size']
    lo, hi = shard_range(T, rank, world)
    train, valid = SyntheticTasks(dataset, p['ways'], p['shots'], 0), SyntheticTasks(dataset, p['ways'], p['shots'], 10 ** 6)
    metrics = {}
    zero = torch.zeros((), device=device)
    save_dir = p.get('save_dir') or ''
    if save_dir and rank == 0:
        os.makedirs(os.path.join(save_dir, 'model_checkpoints'), exist_ok=True)
    for it in range(p['num_iterations']):
        opt.zero_grad()
        ids = list(range(it * T + lo, it * T + hi))
        # BatchNorm buffers (saved with every checkpoint, utils/experiment.py:85-90): tracked when checkpoints are written
        fold = RunningStatsFold(model.engine(), model.base, model.spec(), T, lo, hi, p['adapt_steps'] + 1,
                                p['ways'] * p['shots']) if save_dir else None
        if ids and not fold:
            # train and validation halves of the iteration (maml_vision.py:102-124) in ONE fused call: the first len(ids) tasks carry
            # the backward half, the validation tasks ride through the same launches
            d, l = train.sample_batch(ids)
            dv, lv = valid.sample_batch([10 ** 6 + i for i in ids])
            n = len(ids)
            total, losses, accs = meta_batch_adapt(maml.clone(), torch.cat([d, dv]).to(device), torch.cat([l, lv]).to(device),
                                                   p['adapt_steps'], p['shots'], p['ways'], grad_tasks=n)
            total.backward()                                                 # accumulates the SUM over this rank's train tasks
            sums = [losses[:n].sum(), accs[:n].sum(), losses[n:].sum(), accs[n:].sum()]
        elif ids:                                # with checkpoints: two calls, so that the BatchNorm buffers fold in the reference's call order
            d, l = train.sample_batch(ids)
            total, losses, accs = meta_batch_adapt(maml.clone(), d.to(device), l.to(device), p['adapt_steps'], p['shots'], p['ways'])
            total.backward()                                                 # accumulates the SUM over this rank's tasks
            fold.collect(0)
            with torch.no_grad():
                d, l = valid.sample_batch([10 ** 6 + i for i in ids])
                _, vlosses, vaccs = meta_batch_adapt(maml.clone(), d.to(device), l.to(device), p['adapt_steps'], p['shots'], p['ways'])
            fold.collect(1)
            sums = [losses.sum(), accs.sum(), vlosses.sum(), vaccs.sum()]
        else:                                    # meta_batch_size < world size: this rank owns no task, contributes zeros
            sums = [zero, zero, zero, zero]
        flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in maml.parameters()])
        flat, lsum, asum, ex = reduce_meta_batch(flat, sums[0], sums[1], extra=sums[2:] + ([fold.contribution] if fold else []))
        vlsum, vasum = ex[0], ex[1]                                           # one all-reduce, valid sums (and buffers) included
        if fold:
            fold.apply(ex[2])
        off = 0
        for q in maml.parameters():                                          # maml_vision.py:139-140
            g = flat[off:off + q.numel()].view_as(q) * (1.0 / T)
            q.grad = g.clone() if q.grad is None else q.grad.copy_(g)
            off += q.numel()
        opt.step()
        metrics = {'train_loss': (lsum / T).item(), 'train_acc': (asum / T).item(),
                   'valid_loss': (vlsum / T).item(), 'valid_acc': (vasum / T).item()}
        if rank == 0:
            log(f'iter {it}: {metrics}')
            if save_dir and it % p['save_every'] == 0:                        # maml_vision.py:143-144, utils/experiment.py:85-90
                torch.save(model.state_dict(), os.path.join(save_dir, 'model_checkpoints', f'model_{it}.pt'))
    test = SyntheticTasks(dataset, p['ways'], p['shots'], 2 * 10 ** 6)
    metrics['test_acc'] = evaluate(p, test, maml, loss, device)
    if world > 1:
        torch.distributed.destroy_process_group()
    if save_dir and rank == 0:
        torch.save(model.state_dict(), os.path.join(save_dir, 'model.pt'))       # utils/experiment.py:85-87
    return model, metrics


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='MAML on Vision (MI355X engine)')
    parser.add_argument('--dataset', type=str, default='min', help='omni or min')
    for k, v in params.items():
        parser.add_argument(f'--{k}', type=type(v), default=v)
    parser.add_argument('--first_order', action='store_true')
    parser.add_argument('--save_dir', type=str, default='', help='write model_checkpoints/model_<it>.pt every save_every iterations and model.pt at the end (reference state_dict keys)')
    args = parser.parse_args()
    for k in params:
        params[k] = getattr(args, k)
    params['save_dir'] = args.save_dir
    run(args.dataset, params, first_order=args.first_order)
