#!/usr/bin/env python3
"""Where a launch of the one-launch tail (csrc/tail.hip) spends its time: the 100 MHz wall clock at its stage boundaries (mi_debug_tail_stamps),
thread 0 of each of a task's four workgroups, for the LAST tail launch of a meta-batch call (first-order call: the query pass's primal tail;
second-order call: the last Hessian-vector pass's tangent tail).

    python tools/tail_stamps.py [--tasks 1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402

NAMES = ['start', 'weights / features staged', 'BatchNorm + pool done (barrier)', 'rows done (barrier)', 'row scalars written through',
         'dWl partial stored', 'df + BatchNorm terms', 'partial stored', 'arrived (drain + atomic)', 'fold done']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tasks', type=int, default=1)
    a = ap.parse_args()
    wl = bench.WORKLOADS['cfg2']
    spec = ModelSpec.mini_imagenet(wl['ways'])
    eng = MetaEngine(spec)
    theta = bench.init_theta(spec).cuda()
    T = a.tasks
    data, labels = synthetic.make_meta_batch('min', list(range(T)), wl['ways'], wl['shots'])
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    buf = torch.zeros(T * 4 * 16, dtype=torch.int64, device='cuda')
    for fo, what in ((True, 'primal tail (query pass of a first-order call)'), (False, 'tangent tail (last Hessian-vector pass)')):
        for _ in range(3):
            eng.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=fo)
        torch.cuda.synchronize()
        eng.lib.mi_debug_tail_stamps(eng._h, C.c_void_p(buf.data_ptr()))
        eng.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=fo)
        torch.cuda.synchronize()
        eng.lib.mi_debug_tail_stamps(eng._h, None)
        s = buf.cpu().numpy().reshape(T, 4, 16).astype(np.int64)
        t0 = s[0, :, 0].min()
        print(f'# {what}, {T} task(s) per call; microseconds since the first workgroup started (100 MHz clock); task 0')
        for g in range(4):
            row = s[0, g, :10]
            print(f'workgroup {g}: ' + '  '.join(f'{(v - t0) / 100.0:6.2f}' if v >= t0 else '     -' for v in row))      # (an older launch's last arriver may have been another workgroup: its stamp is stale)
        print('stages: ' + ' | '.join(f'{i}={n}' for i, n in enumerate(NAMES)))
        buf.zero_()


if __name__ == '__main__':
    main()
