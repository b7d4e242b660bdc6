#!/usr/bin/env python3
"""fp64 margins of the pooling / ReLU decisions of one task's passes (CPU, oracle arithmetic): per block the smallest gap between a
pooling window's maximum and its runner-up and the smallest |u| at a window's maximum.  A margin at the level of one fp32 rounding
(~1e-7 of the activations' scale) is a decision fp32 arithmetic can resolve either way -- the engine's two operand forms of the hidden
convolutions, and the reference's own fp32 run, then differ on that task by ~1e-3..1e-2 in the gradient while agreeing to 1e-6 everywhere
else.  The tests' plateau-free task 705 (tests/test_gpu_engine.py) has one at 7e-7 in block 3 of its support pass.

    python tools/decision_margins.py [task seed offsets ...]        (default: 5 0)"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from exploring_meta_amd.utils import synthetic  # noqa: E402
from helpers import model_params  # noqa: E402
from oracle import vision_ref as R  # noqa: E402

spec = R.mini_imagenet_spec(5)
theta = {k: v.double() for k, v in model_params(spec, 5).items()}
for t in [int(a) for a in sys.argv[1:]] or [5, 0]:
    data = torch.from_numpy(synthetic.hash_uniform(700 + t, (10, 3, 84, 84)) * 255.0).double()
    for half in (0, 1):
        acts = data[half::2]
        for blk in range(4):
            w, b = theta[f'base.{blk}.conv.weight'], theta[f'base.{blk}.conv.bias']
            g, be = theta[f'base.{blk}.normalize.weight'], theta[f'base.{blk}.normalize.bias']
            u = F.batch_norm(F.conv2d(acts, w, b, padding=1), None, None, g, be, True, 0.0, 1e-5)
            n, c, h, wd = u.shape
            hp, wp = h // 2, wd // 2
            win = u[:, :, :hp * 2, :wp * 2].reshape(n, c, hp, 2, wp, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, hp, wp, 4)
            top2 = win.sort(dim=-1, descending=True).values[..., :2]
            gap = (top2[..., 0] - torch.clamp(top2[..., 1], min=0.0))[top2[..., 0] > 0]
            print(f'task {700 + t} half {half} block {blk + 1}: smallest argmax gap {float(gap.min()):.3e}, smallest |max u| '
                  f'{float(top2[..., 0].abs().min()):.3e} (u scale {float(u.abs().max()):.2f}, {win.shape[:-1].numel()} windows)')
            acts = F.max_pool2d(F.relu(u), 2)
