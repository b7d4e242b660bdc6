#!/usr/bin/env python3
"""Probe: is there anything to win by running the meta-batch as two independent half-batches on two streams (HBM-bound BatchNorm
kernels of one half under the matrix-bound convs of the other)?  Two engines, 16 tasks each, on two torch streams, against one
engine with 32 tasks.  (Answer recorded in DESIGN.md 8b.)"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402


def main():
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
    spec = ModelSpec.mini_imagenet(wl['ways'])
    theta = bench.init_theta(spec).cuda()
    data, labels = bench.make_batch(wl, list(range(32)))
    data, labels = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    halves = [(data[:16].contiguous(), labels[:16].contiguous()), (data[16:].contiguous(), labels[16:].contiguous())]
    one = MetaEngine(spec)
    two = [MetaEngine(spec), MetaEngine(spec)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run_one():
        one.meta_batch(theta, data, labels, wl['shots'], wl['steps'], wl['lr'])

    def run_two():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for e, s, (d, l) in zip(two, streams, halves):
            with torch.cuda.stream(s):
                e.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr'])
        for s in streams:
            cur.wait_stream(s)

    for name, fn in (('one engine, 32 tasks', run_one), ('two engines x 16 tasks on two streams', run_two), ('one engine, 32 tasks', run_one)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20 if wl['steps'] > 1 else 200
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f'{name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per 32 tasks', flush=True)


if __name__ == '__main__':
    main()
