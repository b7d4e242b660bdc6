#!/usr/bin/env python3
"""Stage timing of the fused Fisher-vector-product sweep (csrc/policy_sweep.h) from in-kernel shader-clock stamps of workgroup 0
(debug aid; cfg5 size)."""
import ctypes as C
import os
import sys
from copy import deepcopy

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from exploring_meta_amd import _lib, core_functions as cf  # noqa: E402
from exploring_meta_amd.core_functions import rl as prl  # noqa: E402

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
ctx = prl._SurrogateContext(replays, olds, policy, baseline, p)
ctx.evaluate(theta, want_grad=True)
v = torch.randn_like(theta)
for _ in range(3):
    ctx.fvp(theta, v)
lib = _lib.load()
buf = torch.zeros(256, dtype=torch.int64, device='cuda')
lib.mi_debug_policy_sweep_stamps(C.c_void_p(buf.data_ptr()))
ctx.fvp(theta, v)
torch.cuda.synchronize()
lib.mi_debug_policy_sweep_stamps(None)
st = buf.cpu().numpy().astype(np.uint64)
names = {0: 'start', 1: 'weights/flush done', 2: 'staged', 3: 'h1d done', 4: 'tangent fwd (MFMA) done', 5: 'mud done', 6: 'gauss done', 7: 'dW3 done',
         8: 'r2 done', 9: 'db2 + r1 (MFMA) done', 10: 'dW2 (MFMA) done', 11: 'loop end', 12: 'flushed'}
prev = None
for x in st:
    x = int(x)
    if x == 0:
        break
    k, t = x >> 56, x & ((1 << 56) - 1)
    print(f'{names.get(k, k):28s} +{(t - prev) if prev is not None else 0:8d} cycles')
    prev = t
