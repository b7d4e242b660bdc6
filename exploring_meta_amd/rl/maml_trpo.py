#!/usr/bin/env python3
"""Counterpart of the reference's ``rl/maml_trpo.py`` for Particles2D on the batched HIP policy engine.

Same structure (rl/maml_trpo.py:82-134): per iteration sample ``meta_batch_size`` tasks, for each task
``learner = deepcopy(policy)`` -> ``fast_adapt_trpo(task, learner, baseline, params, first_order=True)``, then
``meta_optimize_trpo(params, policy, baseline, iter_replays, iter_policies)``.  The environment (learn2learn Particles2D) and
the cherry Runner are replaced by ``Particles2DRunner``.  With torchrun the task list is sharded over ranks and the meta
optimisation's means are completed by small all-reduces (core_functions/rl.py::_SurrogateContext._allmean).

    python -m exploring_meta_amd.rl.maml_trpo --meta_batch_size 20 --num_iterations 5
"""
import argparse
import os
import random
from copy import deepcopy

import numpy as np
import torch

from ..core_functions import (DiagNormalPolicy, DiagNormalPolicyANIL, LinearValue, Particles2DRunner, fast_adapt_trpo, meta_optimize_trpo,
                              set_device)
from ..sharding import init_process_group, shard_range

params = {
    'inner_lr': 0.1, 'max_path_length': 100, 'adapt_steps': 1, 'adapt_batch_size': 20, 'meta_batch_size': 20,
    'outer_lr': 0.3, 'backtrack_factor': 0.5, 'ls_max_steps': 15, 'max_kl': 0.01, 'tau': 1.0, 'gamma': 0.99,
    'num_iterations': 10, 'seed': 42,
}


def run(p, log=print, anil=False):
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    if world > 1:
        init_process_group(local)
    dev = torch.device('cuda', local)
    set_device(dev)
    random.seed(p['seed']); np.random.seed(p['seed']); torch.manual_seed(p['seed'])
    rng = np.random.RandomState(p['seed'])
    gen = torch.Generator(device=dev).manual_seed(p['seed'] + rank)
    baseline = LinearValue(2, 2)                      # reference passes env.action_size as the ridge coefficient (:85)
    # anil: rl/anil_trpo.py:85 -- tanh body + linear head, inner updates with the body under no_grad (rl.py:381-382)
    policy = (DiagNormalPolicyANIL(2, 2, p.get('fc_neurons', 100)) if anil else DiagNormalPolicy(2, 2)).to(dev)
    lo, hi = shard_range(p['meta_batch_size'], rank, world)
    for it in range(p['num_iterations']):
        goals = rng.uniform(-0.5, 0.5, size=(p['meta_batch_size'], 2))        # env.sample_tasks: identical on every rank
        iter_replays, iter_policies, iter_reward, iter_loss = [], [], 0.0, 0.0
        for goal in goals[lo:hi]:
            learner = deepcopy(policy)
            task = Particles2DRunner(goal, p['max_path_length'], gen, dev)
            learner, eval_loss, task_replay, task_rew, _ = fast_adapt_trpo(task, learner, baseline, p, anil=anil, first_order=True)
            iter_reward += task_rew
            iter_loss += eval_loss.item()
            iter_replays.append(task_replay)
            iter_policies.append(learner)
        out = meta_optimize_trpo(p, policy, baseline, iter_replays, iter_policies, anil=anil)
        if rank == 0:
            log(f'iter {it}: average_return {iter_reward / (hi - lo):.3f} loss {iter_loss / (hi - lo):.4f} '
                f'line-search step {out["accepted"]}')
    if world > 1:
        torch.distributed.destroy_process_group()
    return policy


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='MAML-TRPO on Particles2D (MI355X engine)')
    for k, v in params.items():
        parser.add_argument(f'--{k}', type=type(v), default=v)
    parser.add_argument('--anil', action='store_true', help='ANIL-TRPO (reference rl/anil_trpo.py): DiagNormalPolicyANIL, head-only inner loop')
    args = parser.parse_args()
    for k in params:
        params[k] = getattr(args, k)
    run(params, anil=args.anil)
