#!/usr/bin/env python3
"""Counterpart of the reference's ``vision/anil_vision.py`` on the batched HIP engine.

Same experiment: ``features = Sequential(ConvBase(...), Lambda(view(-1, fc_neurons)))``, ``head = MAML(Linear(fc_neurons,
ways), lr)``, Adam over trunk + head, CrossEntropy, per-iteration train + validation meta-batches, gradient /
meta_batch_size (anil_vision.py:86-99,110-141) -- with the ``for task in range(meta_batch_size)`` loop (:114-132) as ONE
``meta_batch_adapt_anil`` call per half (mi_meta_batch_anil: trunk once on all rows, head-only inner loop, outer gradient into
trunk and head), task-sharded over ranks under ``torchrun`` with one RCCL all-reduce.  Tasks come from the seeded synthetic
generator (datasets are not available offline).

    python -m exploring_meta_amd.vision.anil_vision --dataset min --shots 5 --adapt_steps 1 --num_iterations 10
"""
import argparse
import os
import random

import numpy as np
import torch

from ..core_functions import MAML, ConvBase
from ..core_functions.anil import anil_engine, meta_batch_adapt_anil
from ..core_functions.vision_models import RunningStatsFold
from ..sharding import init_process_group, reduce_meta_batch, shard_range
from .maml_vision import SyntheticTasks

params = {
    "ways": 5, "shots": 5, "outer_lr": 0.003, "inner_lr": 0.5, "adapt_steps": 1, "meta_batch_size": 32,
    "num_iterations": 10000, "save_every": 1000, "seed": 42,
}


class Lambda(torch.nn.Module):
    """reference anil_vision.py:46-53"""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


def build(dataset, ways, inner_lr, device):
    """features / head exactly as the reference constructs them (anil_vision.py:40-43,86-94)."""
    if dataset == 'omni':
        fc_neurons = 128
        features = ConvBase(output_size=64, hidden=32, channels=1, max_pool=False)
    else:
        fc_neurons = 1600
        features = ConvBase(output_size=64, channels=3, max_pool=True)
    features = torch.nn.Sequential(features, Lambda(lambda x: x.view(-1, fc_neurons))).to(device)
    head = MAML(torch.nn.Linear(fc_neurons, ways), lr=inner_lr).to(device)
    return features, head


def run(dataset, p, log=print):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        init_process_group(local)
    random.seed(p['seed']); np.random.seed(p['seed']); torch.manual_seed(p['seed']); torch.cuda.manual_seed(p['seed'])
    device = torch.device('cuda', local)
    features, head = build(dataset, p['ways'], p['inner_lr'], device)
    all_parameters = list(features.parameters()) + list(head.parameters())           # anil_vision.py:97
    opt = torch.optim.Adam(all_parameters, lr=p['outer_lr'])
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
        # BatchNorm buffers of `features` (saved with every checkpoint): one features(data) pass per task and phase (anil.py fast_adapt)
        fold = None
        if save_dir:
            eng = anil_engine(features, p['ways'], 28 if dataset == 'omni' else 84, device)
            fold = RunningStatsFold(eng, features[0], eng.spec, T, lo, hi, 1, 2 * p['ways'] * p['shots'])
        if ids:
            d, l = train.sample_batch(ids)
            total, losses, accs = meta_batch_adapt_anil(head.clone(), features, d.to(device), l.to(device), p['adapt_steps'],
                                                        p['shots'], p['ways'])
            total.backward()
            if fold:
                fold.collect(0)
            with torch.no_grad():
                d, l = valid.sample_batch([10 ** 6 + i for i in ids])
                _, vlosses, vaccs = meta_batch_adapt_anil(head.clone(), features, d.to(device), l.to(device), p['adapt_steps'],
                                                          p['shots'], p['ways'])
            if fold:
                fold.collect(1)
            sums = [losses.sum(), accs.sum(), vlosses.sum(), vaccs.sum()]
        else:                                    # meta_batch_size < world size: this rank owns no task, contributes zeros
            sums = [zero, zero, zero, zero]
        flat = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1) for q in all_parameters])
        flat, lsum, asum, ex = reduce_meta_batch(flat, sums[0], sums[1], extra=sums[2:] + ([fold.contribution] if fold else []))
        vlsum, vasum = ex[0], ex[1]                                           # one all-reduce, valid sums (and buffers) included
        if fold:
            fold.apply(ex[2])
        off = 0
        for q in all_parameters:                                                     # anil_vision.py:139-140
            g = flat[off:off + q.numel()].view_as(q) * (1.0 / T)
            q.grad = g.clone() if q.grad is None else q.grad.copy_(g)
            off += q.numel()
        opt.step()
        metrics = {'train_loss': (lsum / T).item(), 'train_acc': (asum / T).item(),
                   'valid_loss': (vlsum / T).item(), 'valid_acc': (vasum / T).item()}
        if rank == 0:
            log(f'iter {it}: {metrics}')
            if save_dir and it % p['save_every'] == 0:                                # anil_vision.py:143-145
                # save_model_checkpoint(m, 'features_<it+1>') -> model_checkpoints/model_features_<it+1>.pt (utils/experiment.py:89-90)
                torch.save(features.state_dict(), os.path.join(save_dir, 'model_checkpoints', f'model_features_{it + 1}.pt'))
                torch.save(head.state_dict(), os.path.join(save_dir, 'model_checkpoints', f'model_head_{it + 1}.pt'))
    if rank == 0 and save_dir:                                                        # anil_vision.py:163-164
        torch.save(features.state_dict(), os.path.join(save_dir, 'features.pt'))
        torch.save(head.state_dict(), os.path.join(save_dir, 'head.pt'))
    if world > 1:
        torch.distributed.destroy_process_group()
    return (features, head), metrics


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='ANIL on Vision (MI355X engine)')
    parser.add_argument('--dataset', type=str, default='min', help='omni or min')
    for k, v in params.items():
        parser.add_argument(f'--{k}', type=type(v), default=v)
    parser.add_argument('--save_dir', type=str, default='', help='write model_checkpoints/model_{features,head}_<it+1>.pt every save_every iterations and features.pt / head.pt at the end')
    args = parser.parse_args()
    for k in params:
        params[k] = getattr(args, k)
    params['save_dir'] = args.save_dir
    run(args.dataset, params)
