#!/usr/bin/env python3
"""The launches of ONE cfg5 meta-optimisation step in order, from a `rocprofv3 --kernel-trace` of `tools/trpo_step_timing.py --whole-only`
(steps are delimited by the one gae_kernel launch each surrogate context makes): per launch its start inside the step, its duration and the
idle gap in front of it; then totals per phase (context = up to the first surrogate sweep, Fisher-vector products, the rest).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/trpo_step_timing.py --whole-only
    python3 tools/trpo_trace.py DIR > profiles/rN/trpo_step_trace_cfg5.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '')
    n = re.sub(r'at::native::(\(anonymous namespace\)::)?', '', n)
    return n[:90]


def main():
    d = sys.argv[1]
    path = d if os.path.isfile(d) else sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in csv.DictReader(open(path)))
    marks = [i for i, r in enumerate(rows) if r[2].startswith('gae_kernel')]
    if len(marks) < 3:
        sys.exit('fewer than three gae_kernel launches')
    # a step starts a few launches before its gae_kernel (the batch assembly): take the window between the last policy_sweep launch of the
    # previous step and the last launch of this one
    i0, i1 = marks[-2], marks[-1]
    while i0 > 0 and 'policy_sweep' not in rows[i0 - 1][2] and 'mean_tasks' not in rows[i0 - 1][2]:
        i0 -= 1
    while i1 > 0 and 'policy_sweep' not in rows[i1 - 1][2] and 'mean_tasks' not in rows[i1 - 1][2]:
        i1 -= 1
    seg = rows[i0:i1]
    t0 = seg[0][0]
    print(f'# {os.path.basename(path)}: one meta-optimisation step = {len(seg)} launches, {(seg[-1][1] - t0) / 1e3:.1f} us from the first launch to the end of the last')
    print('index,start_us,duration_us,gap_before_us,kernel')
    prev_end = t0
    tot = defaultdict(lambda: [0, 0.0])
    busy = gaps = 0.0
    for i, (s, e, n) in enumerate(seg):
        gap = max(0, s - prev_end) / 1e3
        print(f'{i},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},"{n}"')
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e3
        busy += (e - s) / 1e3
        gaps += gap
        prev_end = max(prev_end, e)
    print(f'#\n# kernel time {busy:.1f} us, idle between launches {gaps:.1f} us')
    print('# kernels by total time: launches, total us, mean us')
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f'# {c:4d} {t:9.1f} {t / c:8.1f}  {n}')


if __name__ == '__main__':
    main()
