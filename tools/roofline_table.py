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

PEAK_TFLOPS = 157.3      # fp32-input MFMA == fp32 vector peak, MI355X_MICROARCH.md
PEAK_GBPS = 8000.0       # HBM3E spec; about 6300 GB/s is achievable with a streaming copy

H = [84, 42, 21, 10]
HP = [42, 21, 10, 5]
CI = [3, 32, 32, 32]
CO = 32


def costs(op, l, images):
    """(flops, bytes) of one launch of `op` on block l over `images` images."""
    x = H[l] * H[l] * CI[l] * 4
    z = H[l] * H[l] * CO * 4
    p = HP[l] * HP[l] * CO * 4
    f = 2 * 9 * CI[l] * CO * H[l] * H[l]
    if l == 0:      # block 1: conv-recompute kernels (z1 / dz1 stay in registers), Gram-matrix statistics, pooled-resolution reductions
        a = p // 4  # argmax bytes (one per pooled element)
        t = {
            'conv_fwd_stats': (f, x), 'bn_relu_pool_fwd': (f, x + 2 * p + a), 'bn_bwd_reduce': (0, 3 * p), 'wgrad': (f, x + p + a),
            'tangent_conv_fwd': (2 * f, x), 'bn_tangent_fwd': (2 * f, x + 2 * p), 'bn_tangent_bwd_reduce': (0, 5 * p),
            'tangent_wgrad': (2 * f, x + 2 * p + a),
        }
    else:
        t = {
            'conv_fwd_stats': (f, x + z), 'dgrad': (f, z + x), 'wgrad': (f, x + z),
            'tangent_conv_fwd': (2 * f, 2 * x + 2 * z), 'tangent_dgrad': (2 * f, 2 * z + x), 'tangent_wgrad': (2 * f, 2 * x + 2 * z),
            'bn_relu_pool_fwd': (0, z + p), 'bn_bwd_reduce': (0, z + p), 'bn_bwd_apply': (0, 2 * z + p),
            'bn_tangent_fwd': (0, 2 * z + p), 'bn_tangent_bwd_reduce': (0, 2 * z + 2 * p), 'bn_tangent_bwd_apply': (0, 3 * z + 2 * p),
        }
    if op not in t:
        return None
    fl, by = t[op]
    return fl * images, by * images


def main():
    path = sys.argv[1]
    images = int(sys.argv[2]) if len(sys.argv) > 2 else 32 * 25
    rows = [r for r in csv.reader(open(path)) if r and not r[0].startswith('#')]
    hdr, rows = rows[0], rows[1:]
    total = sum(float(r[3]) for r in rows)
    print(f'Per-kernel roofline, {path} ({images} images per launch, one meta-iteration = {total:.2f} ms of kernel time)\n')
    print('| op | block | launches | avg ms | share | GFLOP/launch | MB/launch | TFLOP/s (% of 157.3) | GB/s (% of 8000) | bound |')
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
        if l == 0:
            bound = 'hbm' if not fl else ('mfma (sparse)' if 'wgrad' in op else 'valu/latency')
        else:
            bound = 'mfma' if fl and (fl / by) > PEAK_TFLOPS * 1e12 / (PEAK_GBPS * 1e9) else 'hbm'
        tfs = f'{tf:.1f} ({tf / PEAK_TFLOPS * 100:.0f} %)' if fl else '-'
        print(f'| {op} | {l + 1} | {n} | {avg:.4f} | {share * 100:.1f} % | {fl / 1e9:.2f} | {by / 1e6:.1f} | '
              f'{tfs} | {gb:.0f} ({gb / PEAK_GBPS * 100:.0f} %) | {bound} |')


if __name__ == '__main__':
    main()
