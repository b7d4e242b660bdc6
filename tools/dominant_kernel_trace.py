#!/usr/bin/env python3
"""Per-launch extraction of one kernel from a `rocprofv3 --kernel-trace` run, so that bench.py's `roofline.frac` can be re-derived from
files under profiles/ alone (the --stats summary aggregates every launch of a template instantiation: blocks 2, 3 and 4 of the net share one
name and differ only in their grid).

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py --no-dist --no-clock --no-other --no-sampled --no-cpu-baseline --no-overlap ...
    python3 tools/dominant_kernel_trace.py DIR --kernel 'conv3x3_s1_mfma_kernel<32, 2, 2, 0, true, false>' [--bench bench.json] > profiles/rN/rocprofv3_dominant_kernel_*.txt

Launches are grouped by grid shape (x = threads, y = tasks, z) and, with --cycle N, by their position in the repeating launch order
(the two-term tangent convolution runs once per hidden block and Hessian-vector pass: blocks 2, 3, 4, 2, 3, 4, ... so --cycle 3; blocks
2 and 3 take the same one-round grid and differ only in their duration).  The group with the largest total time is the block the
roofline record names (block 2: the 42x42 map); `--bench` adds the algorithmic FLOPs / bytes per launch of the bench line and the
fraction they give."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import OrderedDict


def find_trace(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))
    if not hits:
        sys.exit(f'no *kernel_trace.csv under {path}')
    return hits[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace', help='kernel_trace.csv or the rocprofv3 output directory')
    ap.add_argument('--kernel', required=True, help='substring of the kernel name (as rocprofv3 prints the template instantiation)')
    ap.add_argument('--bench', default='', help="bench.py's JSON line of the un-profiled run (roofline record) for the FLOP / byte counts")
    ap.add_argument('--cycle', type=int, default=1, help='launches of this kernel repeat in a cycle of this length (one per block): group by position too')
    ap.add_argument('--list', type=int, default=12, help='individual launches to list for the dominant group')
    a = ap.parse_args()
    rows = []
    with open(find_trace(a.trace), newline='') as f:
        for row in csv.DictReader(f):
            if a.kernel not in row['Kernel_Name']:
                continue
            rows.append((int(row['Start_Timestamp']), (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6,
                         tuple(int(row[f'Grid_Size_{c}']) for c in 'XYZ') + (int(row.get('Workgroup_Size_X', 0)),)))
    rows.sort()
    # (calls with another task count -- bench.py's untimed comparison call -- would break the cycle: keep the commonest task count)
    if a.cycle > 1 and rows:
        ys = [r[2][1] for r in rows]
        common = max(set(ys), key=ys.count)
        rows = [r for r in rows if r[2][1] == common]
    groups = OrderedDict()
    for i, (t0, d, grid) in enumerate(rows):
        groups.setdefault(grid + ((i % a.cycle,) if a.cycle > 1 else ()), []).append((t0, d))
    if not groups:
        sys.exit(f'no launch of a kernel matching {a.kernel!r}')
    print(f'# kernel: {a.kernel}')
    print(f'# source: rocprofv3 --kernel-trace ({os.path.basename(find_trace(a.trace))}); durations = End_Timestamp - Start_Timestamp of each dispatch')
    print('grid_x,grid_y,grid_z,workgroup' + (',position_in_cycle' if a.cycle > 1 else '') + ',launches,avg_ms,min_ms,max_ms,total_ms')
    dom = max(groups, key=lambda k: sum(d for _, d in groups[k]))
    for k, v in groups.items():
        ds = [d for _, d in v]
        print(','.join(str(x) for x in k) + f',{len(ds)},{sum(ds) / len(ds):.4f},{min(ds):.4f},{max(ds):.4f},{sum(ds):.3f}' + ('   <- dominant group' if k == dom else ''))
    ds = [d for _, d in sorted(groups[dom])]
    print(f'# dominant group, launch by launch (ms, in start order; first {a.list}): ' + ' '.join(f'{d:.4f}' for d in ds[:a.list]))
    avg = sum(ds) / len(ds)
    if a.bench:
        line = [l for l in open(a.bench).read().splitlines() if l.startswith('{')][-1]
        r = json.loads(line)['roofline']
        print(f"# bench.py roofline record (un-profiled run, HIP events): kernel = {r['kernel']}; avg_launch_ms = {r['avg_launch_ms']}, frac = {r['frac']}")
        if r.get('bound') == 'mfma':
            ach = r['flops_per_launch'] / (avg * 1e-3) / 1e12
            print(f"# from this trace: {r['flops_per_launch'] / 1e9:.2f} GFLOP per launch / {avg:.4f} ms = {ach:.1f} TFLOP/s = {ach / r['peak']:.4f} of {r['peak']} {r['unit']}")
        else:
            ach = r['algorithmic_bytes_per_launch'] / (avg * 1e-3) / 1e9
            print(f"# from this trace: {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB per launch / {avg:.4f} ms = {ach:.0f} GB/s = {ach / r['peak']:.4f} of {r['peak']} {r['unit']}")


if __name__ == '__main__':
    main()
