#!/usr/bin/env python3
"""Which parts of a task's inner-step gradient / Hessian-vector product depend on how many tasks share the launch?  The same 5 tasks as
tests/test_gpu_engine.py::test_train_and_validation_tasks_in_one_call: tasks 0..2 run as a 3-task call and inside the 5-task call; per traced tensor
and step, the largest difference over the three shared tasks relative to the largest value, overall and by 16 equal ranges of the flat parameter
vector (0 = bit-identical)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402
import bench  # noqa: E402


def main():
    ways, shots, K, lr = 5, 5, 2, 0.4
    spec = ModelSpec.mini_imagenet(ways)
    theta = bench.init_theta(spec).cuda()
    data, labels = synthetic.make_meta_batch('min', [3, 4, 5, 6, 7], ways, shots)
    d, l = torch.from_numpy(data).cuda().contiguous(), torch.from_numpy(labels).cuda().contiguous()
    eng = MetaEngine(spec)
    traces = {}
    for T in (3, 5):
        tr = eng.set_trace(T, K)
        eng.meta_batch(theta, d[:T], l[:T], shots, K, lr)
        torch.cuda.synchronize()
        traces[T] = {k: tr[k].cpu().numpy().copy() for k in ('theta', 'g', 'lam_in', 'hv')}
        eng.set_trace(0)
    for key in ('g', 'hv', 'theta', 'lam_in'):
        a, b = traces[3][key], traces[5][key]
        print(f'-- {key}: shapes {a.shape} / {b.shape}')
        for step in range(a.shape[0]):
            x, y = a[step][:3], b[step][:3]
            # engine layout (csrc/engine.hip): per block [gamma | beta | W | conv bias (no gradient: train-mode BatchNorm cancels it)], then the head's weight and bias
            names, off, ci = [], 0, 3
            for blk in range(4):
                for nm, n in ((f'g{blk + 1}', 32), (f'b{blk + 1}', 32), (f'W{blk + 1}', 9 * ci * 32), (f'cb{blk + 1}', 32)):
                    names.append((nm, off, off + n)); off += n
                ci = 32
            names += [('Wl', off, off + ways * 800), ('bl', off + ways * 800, off + ways * 800 + ways)]
            row = []
            for nm, lo, hi in names:
                xs, ys = x[:, lo:hi], y[:, lo:hi]
                row.append(f'{nm} {float(np.abs(xs - ys).max()) / max(float(np.abs(xs).max()), 1e-30):.0e}')
            print(f'step {step}: overall {float(np.abs(x - y).max()) / max(float(np.abs(x).max()), 1e-30):.2e} | ' + ' '.join(row))

if __name__ == '__main__':
    main()
