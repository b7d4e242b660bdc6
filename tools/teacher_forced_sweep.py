#!/usr/bin/env python3
"""Teacher-forced per-step parity of BASELINE config 2 (32 tasks, 5-way 5-shot, K = 5, lr 0.5, second order) for ALL tasks:
every inner-step gradient and every Hessian-vector product of the fused call against oracle/vision_ref.py evaluated at the
engine's own theta_k, in fp64 and in the reference's fp32.  Writes the per-task, per-step table (markdown) and a JSON file.
Test infrastructure (runs on the GPU box; the oracle legs run in CPU worker processes)."""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402
from oracle import vision_ref as R  # noqa: E402
import teacher_forced as TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tasks', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(REPO, 'gpurun_out', 'teacher_forced_cfg2'))
    ap.add_argument('--workers', type=int, default=0)
    ap.add_argument('--dump-tasks', default='', help='comma list: save the teacher-forcing job of these tasks (npz) next to --out, for offline analysis')
    args = ap.parse_args()
    ways, shots, K, lr, T = 5, 5, 5, 0.5, args.tasks
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    th0 = OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic.ref_init_weights(R.param_shapes(spec), 42).items())
    theta = R.flatten_params(th0).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    eng = MetaEngine(mspec)
    trace = eng.set_trace(T, K)
    loss, acc, grad, _ = eng.meta_batch(theta, torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), shots, K, lr)
    torch.cuda.synchronize()
    for t in [int(x) for x in args.dump_tasks.split(',') if x]:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        np.savez_compressed(f'{args.out}_job_task{t}.npz', t=t, shots=shots, ways=ways, data=data[t], labels=labels[t],
                            **{k: trace[k][:, t].cpu().numpy() for k in ('theta', 'g', 'lam_in', 'hv')})
    t0 = time.time()
    res = TF.teacher_forced_all(trace, data, labels, shots, ways, list(range(T)), workers=args.workers or None)
    dt = time.time() - t0
    f = lambda xs: ' '.join(f'{x:.1e}' for x in xs)
    lines = [f'Teacher-forced per-step parity, cfg2 (T = {T}, 5w5s, K = 5, lr 0.5, second order): relative L2 error of the engine\'s g_k and '
             f'H_k lam against oracle/vision_ref.py at the engine\'s own theta_k (k = 0..4).  Oracle legs: {dt:.0f} s of CPU.', '',
             f'"near-tie adjusted": against the fp64 arithmetic under the decision assignment (ReLU masks, pooling argmaxes) that differs from the fp64 '
             f'one only at decisions whose fp64 margin is below {TF.TAU:g} (tests/teacher_forced.py::explain_step); "flips" = how many such decisions per step (k = 0..4, then the query pass).', '',
             '| task | g vs fp64 | g vs ref fp32 | g near-tie adjusted | HVP vs fp64 | HVP vs ref fp32 | HVP near-tie adjusted | flips | query grad vs fp64 / fp32 / adjusted | loss engine / fp64 / fp32 | acc eq |',
             '|---|---|---|---|---|---|---|---|---|---|---|']
    for r in res:
        t = r['t']
        lines.append(f"| {t} | {f(r['g64'])} | {f(r['g32'])} | {f(r['gx'])} | {f(r['h64'])} | {f(r['h32'])} | {f(r['hx'])} | {' '.join(str(len(x)) for x in r['flips'])} | "
                     f"{r['q64'][2]:.1e} / {r['q32'][2]:.1e} / {r['qx']:.1e} | "
                     f"{float(loss[t]):.7f} / {r['q64'][0]:.7f} / {r['q32'][0]:.7f} | {int(float(acc[t]) == r['q64'][1])}{int(float(acc[t]) == r['q32'][1])} |")
    allv = {k: np.array([r[k] for r in res]) for k in ('g64', 'g32', 'h64', 'h32', 'gx', 'hx')}
    best_g, best_h = np.minimum(allv['g64'], allv['g32']), np.minimum(allv['h64'], allv['h32'])
    lines += ['', '| quantity | median | share of steps > 1e-4 | share > 1e-3 | max |', '|---|---|---|---|---|']
    for name, v in (('g vs fp64', allv['g64']), ('g vs ref fp32', allv['g32']), ('g vs nearer leg', best_g), ('g near-tie adjusted', allv['gx']),
                    ('HVP vs fp64', allv['h64']), ('HVP vs ref fp32', allv['h32']), ('HVP vs nearer leg', best_h), ('HVP near-tie adjusted', allv['hx'])):
        lines.append(f'| {name} | {np.median(v):.1e} | {(v > 1e-4).mean():.3f} | {(v > 1e-3).mean():.3f} | {v.max():.1e} |')
    margins = [abs(fl['margin']) for r in res for step in r['flips'] for fl in step]
    kinds = [f"block {fl['block'] + 1} {fl['kind']}" for r in res for step in r['flips'] for fl in step]
    lines += ['', f'Flipped decisions: {len(margins)} over {len(res) * 6} passes; largest fp64 margin {max(margins) if margins else 0:.1e}; by kind: '
              + ', '.join(f'{k}: {kinds.count(k)}' for k in sorted(set(kinds)))]
    print('\n'.join(lines), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out + '.md', 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    with open(args.out + '.json', 'w') as fh:
        json.dump(dict(results=res, loss=[float(x) for x in loss], acc=[float(x) for x in acc]), fh)


if __name__ == '__main__':
    main()
