#!/usr/bin/env python3
"""Probe for few tasks per call (the per-rank regime of strong scaling): T tasks as C independent chains of T/C tasks on C streams
(C engines), against one engine with T tasks.  Few-task launches underfill the chip and are latency-bound, so independent chains could
overlap where one large chain cannot be split further."""
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
    engines = [MetaEngine(spec) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for T in (2, 4, 8, 16):
        data, labels = bench.make_batch(wl, list(range(T)))
        data, labels = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
        res = {}
        for C in (1, 2, 4):
            if T % C or T // C < 1:
                continue
            n = T // C
            parts = [(data[i * n:(i + 1) * n].contiguous(), labels[i * n:(i + 1) * n].contiguous()) for i in range(C)]

            def run():
                if C == 1:
                    return engines[0].meta_batch(theta, data, labels, wl['shots'], wl['steps'], wl['lr'])
                cur = torch.cuda.current_stream()
                outs = []
                for s in streams[:C]:
                    s.wait_stream(cur)
                for e, s, (d, l) in zip(engines, streams, parts):
                    with torch.cuda.stream(s):
                        outs.append(e.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr']))
                for s in streams[:C]:
                    cur.wait_stream(s)
                return outs

            for _ in range(3):
                run()
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize()
            res[C] = (time.perf_counter() - t0) / reps * 1e3
        print(f'T={T}: ' + ', '.join(f'{C} chain(s) {ms:.3f} ms' for C, ms in res.items()), flush=True)


if __name__ == '__main__':
    main()
