#!/usr/bin/env python3
"""Where a few-task meta-iteration spends its time ("if it cannot be done, commit the trace that shows the remaining floor").
Reads a `rocprofv3 --kernel-trace` of tools/t_sweep.py at one task count and reports, per meta-iteration (iterations are delimited by the
one prepare_batch launch each of them starts with): launches, wall span, time with at least one kernel running, idle time between kernels, the
distribution of kernel durations, and the kernels by total time.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/t_sweep.py --workload cfg2 --tasks 4 --steps 6
    python3 tools/launch_floor.py DIR --tasks 4 > profiles/rN/launch_floor_cfg2_T4.txt"""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--tasks', type=int, default=0)
    ap.add_argument('--mark', default='prepare_batch_kernel', help='a kernel launched exactly once per meta-iteration, at its start')
    a = ap.parse_args()
    path = a.trace if os.path.isfile(a.trace) else sorted(glob.glob(os.path.join(a.trace, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.mark in r[2]]
    if len(marks) < 4:
        sys.exit(f'fewer than four launches of {a.mark}: cannot delimit iterations')
    its = [(marks[k], marks[k + 1]) for k in range(len(marks) - 4, len(marks) - 1)]          # the last three complete iterations
    print(f'# {os.path.basename(path)}: {len(rows)} dispatches, {len(marks)} meta-iterations; the last three complete ones, tasks per call = {a.tasks}')
    print('iteration,launches,wall_us,busy_us,idle_us,sum_of_durations_us,gaps,mean_gap_us,median_kernel_us,kernels_under_10us,kernels_under_5us')
    agg = defaultdict(lambda: [0, 0.0])
    tot = defaultdict(float)
    gap_pairs = defaultdict(lambda: [0, 0.0])
    for n, (i0, i1) in enumerate(its):
        seg = rows[i0:i1]
        wall = (rows[i1][0] - seg[0][0]) / 1e3
        busy, cur_s, cur_e, gaps = 0.0, seg[0][0], seg[0][1], []
        for s, e, _ in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        busy /= 1e3
        durs = sorted((e - s) / 1e3 for s, e, _ in seg)
        print(f'{n},{len(seg)},{wall:.1f},{busy:.1f},{wall - busy:.1f},{sum(durs):.1f},{len(gaps)},{(sum(gaps) / max(1, len(gaps))) / 1e3:.2f},'
              f'{durs[len(durs) // 2]:.1f},{sum(d < 10 for d in durs)},{sum(d < 5 for d in durs)}')
        # which boundaries idle: (kernel before the gap -> kernel after it), summed over the iterations
        cur_e, cur_k = seg[0][1], seg[0][2]
        for s, e, k in seg[1:]:
            if s > cur_e:
                gp = gap_pairs[(cur_k[:60], k[:60])]
                gp[0] += 1
                gp[1] += (s - cur_e) / 1e3
            if e >= cur_e:
                cur_e, cur_k = e, k
        for s, e, k in seg:
            agg[k][0] += 1
            agg[k][1] += (e - s) / 1e3
        for key, v in (('launches', len(seg)), ('wall', wall), ('busy', busy), ('sum', sum(durs)), ('gaps', len(gaps)), ('gap_us', sum(gaps) / 1e3)):
            tot[key] += v / len(its)
    print(f"\n# per iteration (mean of three): {tot['launches']:.0f} launches, wall {tot['wall']:.0f} us, busy {tot['busy']:.0f} us, idle {tot['wall'] - tot['busy']:.0f} us in "
          f"{tot['gaps']:.0f} gaps ({tot['gap_us'] / max(1.0, tot['gaps']):.2f} us each)")
    print('#\n# idle gaps by boundary (per iteration): count, total us, mean us, kernel before -> kernel after')
    for (ka, kb), (c, t) in sorted(gap_pairs.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f'{c / len(its):6.1f} {t / len(its):8.1f} {t / c:7.2f}  {ka} -> {kb}')
    print('#\n# kernels by total time (per iteration): launches, total us, mean us')
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f'{c / len(its):6.1f} {t / len(its):9.1f} {t / c:8.2f}  {k[:110]}')


if __name__ == '__main__':
    main()
