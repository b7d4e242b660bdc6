#!/usr/bin/env python3
"""Where one cfg5 step (meta_optimize_trpo on 20 tasks x 2000-row replays) spends its time: host-side context construction (GAE,
LinearValue fits, padding / upload), the surrogate + gradient call, the 11 Fisher-vector products, the line search."""
import os
import sys
import time
from copy import deepcopy

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from exploring_meta_amd import core_functions as cf  # noqa: E402
from exploring_meta_amd.core_functions import rl as prl  # noqa: E402


def main():
    p = dict(bench.TRPO_PARAMS, meta_batch_size=20)
    dev = torch.device('cuda', 0)
    cf.set_device(dev)
    torch.manual_seed(42)
    policy = cf.DiagNormalPolicy(2, 2).to(dev)
    baseline = cf.LinearValue(2, 2)
    goals = np.random.RandomState(42).uniform(-0.5, 0.5, size=(20, 2))
    gen = torch.Generator(device=dev).manual_seed(42)
    replays, olds = [], []
    for goal in goals:
        task = cf.Particles2DRunner(goal, p['max_path_length'], gen, dev)
        learner, _, rep, _, _ = cf.fast_adapt_trpo(task, deepcopy(policy), baseline, p, first_order=True)
        replays.append(rep)
        olds.append(learner)
    theta = policy.flat().clone()
    sync = torch.cuda.synchronize

    def timed(fn, n=5):
        fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        sync()
        return (time.perf_counter() - t0) / n * 1e3, r

    def whole():
        policy.load_flat(theta)
        return cf.meta_optimize_trpo(p, policy, baseline, replays, olds)
    if '--whole-only' in sys.argv:                # for a kernel trace of the step alone (tools/trpo_trace.py delimits steps by gae_kernel)
        t_all, _ = timed(whole, 6)
        print(f'whole step {t_all:.2f} ms')
        return
    t_ctx, ctx = timed(lambda: prl._SurrogateContext(replays, olds, policy, baseline, p))
    t_eval, _ = timed(lambda: ctx.evaluate(theta, want_grad=True))
    v = torch.randn_like(theta)
    t_fvp, _ = timed(lambda: ctx.fvp(theta, v), 20)
    t_ls, _ = timed(lambda: ctx.evaluate(theta))
    t_all, _ = timed(whole)
    print(f'context (host GAE / fits / upload) {t_ctx:.2f} ms | surrogate+grad {t_eval:.2f} | fvp {t_fvp:.3f} x 11 = {11 * t_fvp:.2f} | '
          f'line-search evaluation {t_ls:.2f} | whole step {t_all:.2f} ms')


if __name__ == '__main__':
    main()
