#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots: FETCH_SIZE and
WRITE_SIZE do not fit in one pass; on gfx950 FETCH_SIZE counts half of a wide coalesced stream -> x2; WRITE_SIZE is exact; both
are reported in KB).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out>/fetch -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out>/write -- python3 bench.py --workload cfg2 --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py <out>/fetch <out>/write --workload cfg2 --table profiles/r2/pmc_hbm_traffic_cfg2.txt --json profiles/pmc_traffic.json

Launches are grouped by (kernel name, grid size): a kernel's largest grid is block 2 of the classifier (the rows bench.py's
`roofline.traffic` quotes), the next ones blocks 3 and 4.
"""
import argparse
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

# stride-1 conv kernel template arguments <CI, NTERMS, EPI, MODE> -> bench.py / engine op name
CONV_OPS = {('1', '1', '0'): 'conv_fwd_stats', ('1', '0', '1'): 'dgrad', ('1', '3', '1'): 'dgrad',
            ('2', '2', '0'): 'tangent_conv_fwd', ('2', '0', '1'): 'tangent_dgrad', ('2', '3', '1'): 'tangent_dgrad'}


def read_pass(path, counter):
    files = glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        sys.exit(f'no *counter_collection.csv under {path}')
    vals = defaultdict(list)                       # (kernel, grid) -> counter value of every launch
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            vals[(short_name(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    # Two blocks of the classifier can be launched with the SAME grid (one balanced round of resident waves: block 2 with 22 tiles per
    # wave, block 3 with 6): their launches differ 4x in bytes, so a (kernel, grid) group is cut wherever consecutive sorted values
    # differ by more than 1.8x; cluster 0 = the largest.
    acc = {}                                       # (kernel, grid, cluster) -> [launches, sum of counter]
    for (k, g), v in vals.items():
        v = sorted(v, reverse=True)
        c, start = 0, 0
        for i in range(1, len(v) + 1):
            if i == len(v) or (v[i] > 0 and v[i - 1] / v[i] > 1.8) or (v[i] == 0 and v[i - 1] > 0):
                acc[(k, g, c)] = [i - start, sum(v[start:i])]
                c, start = c + 1, i
    return acc


def short_name(n):
    n = re.sub(r'\(.*$', '', n)                    # drop the argument list
    return n.replace('void ', '').strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--workload', default='cfg2')
    ap.add_argument('--table', default='')
    ap.add_argument('--json', default='')
    ap.add_argument('--note', default='')
    ap.add_argument('--pick', action='append', default=[], help="'<kernel regex>::<op>,<block index>': the launches of the matching kernel with the "
                    "LARGEST grid (block 2 of the classifier / the only grid of a policy sweep) give roofline.traffic of that workload's (op, block)")
    args = ap.parse_args()
    fe, wr = read_pass(args.fetch_dir, 'FETCH_SIZE'), read_pass(args.write_dir, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] * 2 + wr.get(k, [0, 0])[1])):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        n = max(nf, nw)
        fetch_mb = 2.0 * sf * 1024 / max(nf, 1) / 1e6        # KB -> bytes, x2 (gfx950 wide-stream correction)
        write_mb = sw * 1024 / max(nw, 1) / 1e6
        rows.append((k[0], k[1], n, fetch_mb, write_mb, k[2]))
    lines = [f'# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --workload {args.workload} --steps 1 --warmup 1 '
             f'--no-cpu-baseline{"; " + args.note if args.note else ""}',
             '# FETCH_SIZE is KB and counts HALF of wide (16 B/lane) coalesced streams on gfx950 (MI355X_MICROARCH.md HBM): shown x2; WRITE_SIZE KB exact',
             '# launches of one kernel with one grid whose byte counts differ by more than 1.8x are listed as separate size classes (0 = largest)',
             'kernel,grid_threads,size_class,launches,fetch_MB_per_launch_x2,write_MB_per_launch']
    lines += [f'"{k}",{g},{c},{n},{f:.1f},{w:.1f}' for k, g, n, f, w, c in rows if n > 0 and (f + w) > 0.5]
    text = '\n'.join(lines) + '\n'
    if args.table:
        open(args.table, 'w').write(text)
    else:
        print(text)
    if args.json:
        out = json.load(open(args.json)) if os.path.exists(args.json) else {}
        out = {k: v for k, v in out.items() if ',' not in k or k.count(',') == 2 or k.startswith('_')}   # drop keys of the old layout
        out['_source'] = ('tools/pmc_traffic.py on rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs); bytes per launch = '
                          '2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts half of 16-B/lane streams, MI355X_MICROARCH.md HBM); '
                          'key = workload,op,block index (0 = block 1)')
        # rows of THIS workload's launches: the grid conv_grid (conv_mfma.hip) gives T tasks of n images at every hidden block
        # (bench.py also runs 16-task comparison calls, which must not be mistaken for deeper blocks)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from exploring_meta_amd.engine import ModelSpec
        from exploring_meta_amd.utils.roofline import layer_geometry
        wl = bench.WORKLOADS[args.workload]
        if wl.get('kind') == 'trpo':
            for spec_ in args.pick:
                rx, key = spec_.split('::')
                cand = [(g, f + w) for k, g, nl, f, w, c in rows if re.search(rx, k) and nl > 0]
                if cand:
                    out[f'{args.workload},{key}'] = int(max(cand)[1] * 1e6)
            json.dump(out, open(args.json, 'w'), indent=1)
            return
        spec = ModelSpec.anil(wl['ways']) if wl.get('anil') else (
            ModelSpec.mini_imagenet(wl['ways']) if wl['dataset'] == 'min' else ModelSpec.omniglot(wl['ways']))
        T, n = wl['tasks'], wl['ways'] * wl['shots'] * (2 if wl.get('anil') else 1)
        geo = layer_geometry(spec)

        def grid_threads(layer, nterms, split_bf16):
            h, w, ci, co, ho, wo = geo[layer][:6]
            ntiles = -(-(n * ho * wo) // (30 if split_bf16 else 32))                                      # conv_grid (csrc/conv_mfma.hip)
            cot = co // 32
            slots = 2048 if split_bf16 else ((1024 if nterms == 2 else 2048) if ci >= 64 else 4096)
            tpw = min(128, max(1, -(-(ntiles * T * cot) // slots)))
            nw = 8 if ((nterms == 2 and ci == 32) or (split_bf16 and ci == 64)) else 4
            return -(-ntiles // (nw * tpw)) * T * cot * nw * 64

        for spec_ in args.pick:
            rx, key = spec_.split('::')
            cand = [(g, -c, f + w) for k, g, nl, f, w, c in rows if re.search(rx, k) and nl > 0]
            if cand:
                out[f'{args.workload},{key}'] = int(max(cand)[2] * 1e6)
        # (a bench run that also times the fp32-pipe form launches both variants of a kernel: the split-bf16 one -- the form the
        # workload's roofline record names -- wins, the other is skipped)
        # (template arguments <CI, NTERMS, EPI, MODE, BF, F16>: BF = a split operand form, F16 = its two-plane fp16 variant)
        base = lambda k: re.sub(r'(, (true|false)){1,2}>$', '', k)
        has_bf = {base(k) for k, g, nl, f, w, c in rows if re.search(r'\d, true(, (true|false))?>$', k) and nl > 0}
        # (the split-bf16 form runs on two kernels since round 5 -- the 16x16x32 one for launches of >= 6 tiles per wave, the 32x32x16 one below
        # that, the same grid rule for both: a layer the first has taken is not a candidate for the second, whose size classes then count
        # through the remaining layers of that grid)
        taken = {}
        top_launches = {(k, g): nl for k, g, nl, f, w, c in rows if c == 0}
        for k, g, nl, f, w, c in sorted(rows, key=lambda r: 0 if 'conv3x3_s1_b16_kernel' in r[0] else 1):
            m = re.match(r'conv3x3_s1_mfma_kernel<(\d+), (\d), (\d), (\d)(?:, (true|false))?(?:, (true|false))?>', k)
            m16 = re.match(r'conv3x3_s1_b16_kernel<(\d+), (\d), (\d), (\d)>', k)      # the split-bf16 form on 16x16x32 MFMAs: same grid rule as BF = true
            if m16:
                m = re.match(r'(\d+), (\d), (\d), (\d), (true)', ', '.join(m16.groups()) + ', true')
            if not (m and (m.group(2), m.group(3), m.group(4)) in CONV_OPS):
                continue
            if m.group(5) == 'false' and base(k) in has_bf:
                continue
            op = CONV_OPS[(m.group(2), m.group(3), m.group(4))]
            same_grid = [layer for layer in range(1, len(geo))       # blocks launched with this grid, largest maps first
                         if geo[layer][2] == int(m.group(1)) and grid_threads(layer, int(m.group(2)), m.group(5) == 'true') == g
                         and (m16 or layer not in taken.get(op, ()))]
            # (a run may also hold calls on FEWER tasks -- bench.py's 16-task comparison leg -- whose launches of a layer form size classes
            # of their own between two layers': the workload's own calls launch every layer equally often, so only classes with as many
            # launches as the largest one count, numbered in order)
            own = sorted(c2 for k2, g2, nl2, f2, w2, c2 in rows if (k2, g2) == (k, g) and nl2 == top_launches[(k, g)])
            if c in own and own.index(c) < len(same_grid):
                layer = same_grid[own.index(c)]
                out[f'{args.workload},{op},{layer}'] = int((f + w) * 1e6)
                if m16:
                    taken.setdefault(op, set()).add(layer)
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
