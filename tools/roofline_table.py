#!/usr/bin/env python3
"""Per-kernel roofline table from a `bench.py --breakdown` CSV (cfg2 geometry: MiniImagenetCNN-32, T tasks x N images).

    python tools/roofline_table.py profiles/r1/event_breakdown_cfg2_v8.csv > profiles/r1/roofline_table_cfg2_v8.md

Algorithmic bytes of one launch = the tensors the op must read or write once (SURVEY.md section 8(d) accounting: layer input,
conv output z, pooled output p; parameters are negligible), algorithmic FLOPs = 2*9*Ci*Co per output pixel per GEMM term.
Block 1 ("layer" 0): forward / tangent-forward are the conv-recompute kernels of block1.hip (FLOPs = the recomputed
convolution, bytes = what reaches HBM); its statistics come from the Gram matrix (gram_stats: latency), its BN-backward
reductions stream pooled tensors (HBM), its weight gradients are the sparse MFMA pass of gram.hip (algorithmic wgrad FLOPs).
"""
import csv
import sys

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd.engine import ModelSpec  # noqa: E402
from exploring_meta_amd.utils.roofline import PEAK_GBPS, mfma_peak, op_costs  # noqa: E402

SPEC = ModelSpec.mini_imagenet(5)


def costs(op, l, images):
    return op_costs(SPEC, op, l, images)


def main():
    args = [a for a in sys.argv[1:] if a not in ('--fp32-pipe', '--fp16')]
    # the operand form the hidden convolutions ran in: three bf16 planes (default), the opt-in two scaled fp16 planes, or the fp32 pipe
    split = 0 if '--fp32-pipe' in sys.argv[1:] else (2 if '--fp16' in sys.argv[1:] else 1)
    path = args[0]
    images = int(args[1]) if len(args) > 1 else 32 * 25
    rows = [r for r in csv.reader(open(path)) if r and not r[0].startswith('#')]
    hdr, rows = rows[0], rows[1:]
    total = sum(float(r[3]) for r in rows)
    print(f'Per-kernel roofline, {path} ({images} images per launch, one meta-iteration = {total:.2f} ms of kernel time)\n')
    print('FLOPs are algorithmic fp32 FLOPs.  Matrix peak per kernel: 157.3 TFLOP/s on the fp32 pipe; 416.7 (= dense bf16 2500 / 6 products per '
          'multiply-add) for the kernels on the split-bf16 operand form (marked bf16x6); 833.3 (/ 3) on the opt-in two-plane fp16 form (fp16x3).\n')
    print('| op | block | launches | avg ms | share | GFLOP/launch | MB/launch | TFLOP/s (% of its matrix peak) | GB/s (% of 8000) | bound |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for r in rows:
        op, l, n, tot, avg, share = r[0], int(r[1]), int(r[2]), float(r[3]), float(r[4]), float(r[5])
        c = costs(op, l, images)
        if c is None:
            print(f'| {op} | {l + 1} | {n} | {avg:.4f} | {share * 100:.1f} % | | | | | latency |')
            continue
        fl, by = c
        tf = fl / (avg * 1e-3) / 1e12
        gb = by / (avg * 1e-3) / 1e9
        peak, pipe = mfma_peak(SPEC, op, l, split)
        if l == 0:
            bound = 'hbm' if not fl else ('mfma (sparse)' if 'wgrad' in op else 'valu/latency')
        else:
            bound = 'mfma' if fl and (fl / by) > peak * 1e12 / (PEAK_GBPS * 1e9) else 'hbm'
        if pipe != 'fp32':
            bound += ' fp16x3' if 'fp16' in pipe else (' bf16x8' if 'x8' in pipe else ' bf16x6')
        tfs = f'{tf:.1f} ({tf / peak * 100:.0f} %)' if fl else '-'
        print(f'| {op} | {l + 1} | {n} | {avg:.4f} | {share * 100:.1f} % | {fl / 1e9:.2f} | {by / 1e6:.1f} | '
              f'{tfs} | {gb:.0f} ({gb / PEAK_GBPS * 100:.0f} %) | {bound} |')


if __name__ == '__main__':
    main()
