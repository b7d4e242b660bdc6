#!/usr/bin/env python3
"""Counterpart of the reference's ``rl/maml_ppo.py`` (and ``rl/anil_ppo.py`` with ``--anil``) for Particles2D on the HIP
policy engine.

Same structure (rl/maml_ppo.py:84-131): ``policy = MAML(DiagNormalPolicy(...), lr=inner_lr)``, Adam(outer_lr) on its
parameters; per iteration and task ``learner = policy.clone()`` -> ``fast_adapt_ppo(task, learner, baseline, params)`` ->
``(eval_loss, reward, success)``; ``av_loss = sum / meta_batch_size``; ``av_loss.backward(); meta_optimizer.step()``.  The
second-order gradient through the ``ppo_epochs`` clipped-surrogate updates comes out of one fused call per task
(mi_policy_meta_batch).  Under torchrun the task list is sharded over ranks and the gradients are all-reduced (RCCL).

    python -m exploring_meta_amd.rl.maml_ppo --meta_batch_size 20 --num_iterations 5 [--anil]
"""
import argparse
import os
import random

import numpy as np
import torch

from ..core_functions import (MAML, DiagNormalPolicy, DiagNormalPolicyANIL, LinearValue, Particles2DRunner, fast_adapt_ppo, set_device)
from ..sharding import init_process_group, shard_range

params = {
    'ppo_epochs': 3, 'ppo_clip_ratio': 0.1, 'inner_lr': 0.01, 'max_path_length': 100, 'adapt_steps': 1, 'adapt_batch_size': 20,
    'meta_batch_size': 20, 'outer_lr': 0.01, 'activation': 'tanh', 'tau': 1.0, 'gamma': 0.99, 'num_iterations': 10, 'seed': 42,
}


def run(p, anil=False, log=print):
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', '1'), ('RANK', '0'), ('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    if world > 1:
        init_process_group(local)
    dev = torch.device('cuda', local)
    set_device(dev)
    random.seed(p['seed']); np.random.seed(p['seed']); torch.manual_seed(p['seed'])
    rng = np.random.RandomState(p['seed'])
    gen = torch.Generator(device=dev).manual_seed(p['seed'] + rank)
    baseline = LinearValue(2, 2)
    net = DiagNormalPolicyANIL(2, 2, 100) if anil else DiagNormalPolicy(2, 2, activation=p['activation'])
    policy = MAML(net.to(dev), lr=p['inner_lr'])
    meta_optimizer = torch.optim.Adam(policy.parameters(), lr=p['outer_lr'])
    T = p['meta_batch_size']
    lo, hi = shard_range(T, rank, world)
    for it in range(p['num_iterations']):
        meta_optimizer.zero_grad()
        goals = rng.uniform(-0.5, 0.5, size=(T, 2))                           # env.sample_tasks: identical on every rank
        iter_reward, iter_loss = 0.0, 0.0
        for goal in goals[lo:hi]:
            learner = policy.clone()
            task = Particles2DRunner(goal, p['max_path_length'], gen, dev)
            eval_loss, task_rew, _ = fast_adapt_ppo(task, learner, baseline, p, anil=anil)
            iter_reward += task_rew
            iter_loss = iter_loss + eval_loss
        av_loss = iter_loss / T                                                # this rank's share of the meta-batch mean
        av_loss.backward()
        if world > 1:
            flat = torch.cat([q.grad.reshape(-1) for q in policy.parameters()] + [av_loss.detach().reshape(1), torch.tensor([iter_reward], device=dev)])
            torch.distributed.all_reduce(flat)
            off = 0
            for q in policy.parameters():
                q.grad.copy_(flat[off:off + q.numel()].view_as(q))
                off += q.numel()
            av_loss, iter_reward = flat[off], flat[off + 1].item()
        meta_optimizer.step()
        if rank == 0:
            log(f'iter {it}: average_return {iter_reward / T:.3f} loss {float(av_loss.detach()):.5f}')
    if world > 1:
        torch.distributed.destroy_process_group()
    return policy


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='MAML-PPO / ANIL-PPO on Particles2D (MI355X engine)')
    for k, v in params.items():
        parser.add_argument(f'--{k}', type=type(v), default=v)
    parser.add_argument('--anil', action='store_true')
    args = parser.parse_args()
    for k in params:
        params[k] = getattr(args, k)
    run(params, anil=args.anil)
