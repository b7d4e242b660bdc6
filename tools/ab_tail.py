#!/usr/bin/env python3
"""A/B of the pass-tail modes (mi_engine_set_fused_tail 0 / 1) on one box: ms per meta-iteration of the train half (engine call +
Adam) at several task counts, modes interleaved round by round so that clock drift hits all alike.  Also checks the meta-gradient is
bit-identical between modes."""
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
    ap.add_argument('--tasks', default='32,4,1')
    ap.add_argument('--modes', default='1,0')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=4)
    args = ap.parse_args()
    wl = bench.WORKLOADS[args.workload]
    spec = ModelSpec.mini_imagenet(wl['ways']) if wl['dataset'] == 'min' else ModelSpec.omniglot(wl['ways'])
    modes = [int(m) for m in args.modes.split(',')]
    theta0 = bench.init_theta(spec).cuda()
    for T in [int(x) for x in args.tasks.split(',')]:
        data, labels = synthetic.make_meta_batch(wl['dataset'], list(range(T)), wl['ways'], wl['shots'])
        d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
        engs, grads = {}, {}
        for m in modes:
            engs[m] = MetaEngine(spec)
            engs[m].set_fused_tail(m)
            grads[m] = engs[m].meta_batch(theta0, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])[2].clone()
        same = all(torch.equal(grads[m], grads[modes[0]]) for m in modes)
        best = {m: 1e9 for m in modes}
        for r in range(args.rounds):
            for m in modes:
                eng, theta, adam = engs[m], theta0.clone(), {}

                def step():
                    loss, acc, grad, _ = eng.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
                    eng.adam_step(theta, grad, adam, 0.003, grad_scale=1.0 / T)

                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                best[m] = min(best[m], (time.perf_counter() - t0) / args.steps * 1e3)
        print(f'T={T}: ' + ', '.join(f'mode {m}: {best[m]:.3f} ms' for m in modes) + f'  (best of {args.rounds} rounds; gradients identical: {same})',
              flush=True)


if __name__ == '__main__':
    main()
