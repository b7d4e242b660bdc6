#!/usr/bin/env python3
"""Debug aid (round 5): where does the 16x16x32 convolution kernel (csrc/conv_b16.h) differ from the fp64 oracle?  Error map by pixel row of the
tile, image row / column and channel parity for one forward convolution through the C ABI (mi_conv3x3_bn_stats).  GPU box only."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from exploring_meta_amd import _lib
from exploring_meta_amd.utils import synthetic
from oracle import kernels_ref as KR
from gpu_utils import dev, ptr, stream

lib = _lib.load()
T, n, h, w, ci, co = 1, 2, 42, 42, 32, 32
if len(sys.argv) > 1:
    n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = synthetic.hash_uniform(10, (T, n, h, w, ci)) * 2.0
w9 = synthetic.hash_uniform(11, (T, 9, ci, co)) * 0.6 - 0.3
mode = os.environ.get('DBG_MODE', '')
if mode == 'center':          # only the centre tap
    w9[:, [0, 1, 2, 3, 5, 6, 7, 8]] = 0
elif mode == 'left':
    w9[:, [0, 1, 2, 4, 5, 6, 7, 8]] = 0
elif mode == 'right':
    w9[:, [0, 1, 2, 3, 4, 6, 7, 8]] = 0
elif mode == 'up':
    w9[:, [0, 2, 3, 4, 5, 6, 7, 8]] = 0
pstride = 9 * ci * co + 17
wbuf = np.zeros((T, pstride), np.float32)
wbuf[:, :9 * ci * co] = w9.reshape(T, -1)
xd, wd = dev(x), dev(wbuf)
for form, b16 in ((1, 1), (1, 0)):
    lib.mi_conv_set_split_bf16(form)
    lib.mi_conv_set_b16(b16)
    z = torch.full((T, n, h, w, co), float('nan'), device='cuda')
    mu = torch.empty(T, co, device='cuda')
    rstd = torch.empty(T, co, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, co)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    _lib.check(lib.mi_conv3x3_bn_stats(stream(), ptr(xd), ptr(wd), pstride, T, n, h, w, ci, co, 1, ptr(z), ptr(mu), ptr(rstd), ptr(scratch), sb))
    torch.cuda.synchronize()
    zr = KR.conv3x3(torch.from_numpy(x.astype(np.float32)).double()[0], torch.from_numpy(w9.astype(np.float32)).double()[0], 1).numpy()
    zz = z[0].cpu().numpy().astype(np.float64)
    err = np.abs(zz - zr)
    scale = np.abs(zr).mean() + 1e-30
    bad = err > 1e-4 * scale
    print(f'form {form} b16 {b16}: rel {np.linalg.norm(zz - zr) / np.linalg.norm(zr):.3e}; bad elements {bad.sum()} of {bad.size}; nan {np.isnan(zz).sum()}')
    if bad.any():
        flat = bad.reshape(-1, co)                       # [pixel][channel]
        pix = np.arange(flat.shape[0])
        pm = (pix + 1) % 30                              # pixel row of the tile: pix = tile*30 - 1 + pm
        pm = np.where(pm == 0, 30, pm)
        print(' bad by tile row pm :', {int(k): int(flat[pm == k].sum()) for k in range(1, 31) if flat[pm == k].sum()})
        col = pix % w
        print(' bad by image column:', {int(k): int(flat[col == k].sum()) for k in range(w) if flat[col == k].sum()})
        print(' bad by channel     :', {int(k): int(flat[:, k].sum()) for k in range(co) if flat[:, k].sum()})
        i = np.argwhere(flat)[:6]
        for p_, c_ in i:
            print(f'   pixel {p_} (img {p_ // (h*w)}, y {(p_ % (h*w)) // w}, x {p_ % w}) ch {c_}: got {zz.reshape(-1, co)[p_, c_]:.6f} want {zr.reshape(-1, co)[p_, c_]:.6f}')
