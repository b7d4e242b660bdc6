#!/usr/bin/env python3
"""Single-GPU sweep over tasks per call (the per-rank proxy for strong scaling: a meta-batch of 32 on N GPUs leaves 32/N tasks per
rank): tasks/s and ms per meta-iteration of the train half (engine call + Adam) for T in a list.  Writes a markdown table."""
import argparse
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg2')
    ap.add_argument('--tasks', default='1,2,4,8,16,32,64')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--out', default='')
    ap.add_argument('--no-overlap', action='store_true', help='weight gradients on the main stream (isolated kernel durations in a trace)')
    args = ap.parse_args()
    wl = bench.WORKLOADS[args.workload]
    spec = ModelSpec.anil(wl['ways']) if wl.get('anil') else (
        ModelSpec.mini_imagenet(wl['ways']) if wl['dataset'] == 'min' else ModelSpec.omniglot(wl['ways']))
    eng = MetaEngine(spec)
    if args.no_overlap:
        eng.set_overlap(False)
    run = eng.meta_batch_anil if wl.get('anil') else eng.meta_batch
    theta = bench.init_theta(spec).cuda()
    rows = []
    for T in [int(x) for x in args.tasks.split(',')]:
        data, labels = synthetic.make_meta_batch(wl['dataset'], list(range(T)), wl['ways'], wl['shots'])
        d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
        adam = {}

        def step():
            loss, acc, grad, _ = run(theta, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
            eng.adam_step(theta, grad, adam, 0.003, grad_scale=1.0 / T)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        rows.append((T, dt * 1e3, T / dt))
        print(f'T={T}: {dt * 1e3:.3f} ms/iter, {T / dt:.1f} tasks/s', flush=True)
    if args.out:
        base = dict((r[0], r[2]) for r in rows)
        with open(args.out, 'w') as f:
            f.write(f'Single-GPU sweep over tasks per call, workload {args.workload} ({wl["name"]}), {args.steps} timed iterations each\n\n')
            f.write('| tasks per call | ms per iteration | tasks/s | vs 32 tasks per call |\n|---|---|---|---|\n')
            for T, ms, tps in rows:
                rel = f'{tps / base[32]:.2f}' if 32 in base else '-'
                f.write(f'| {T} | {ms:.3f} | {tps:.1f} | {rel} |\n')


if __name__ == '__main__':
    main()
