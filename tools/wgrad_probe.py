#!/usr/bin/env python3
"""In-process A/B of the stride-1 weight-gradient kernels (launch + partial fold, HIP events): the fp32 rows kernel vs the split-bf16 strip
and unit forms (mi_conv_set_split_bf16 with a variant mask), at block geometries of the 4-conv-32 classifier, with the relative
difference of each result from the fp32 kernel's.  Box-to-box variance is +-10 %: compare within one run only.  Ablations (no MFMAs /
no operand preparation / no loads) are separate builds: csrc/wgrad_bf16.hip, MI_WGRAD_DBG, selected with MI_MAML_LIB."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd import _lib  # noqa: E402

lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
ALL = 0x3ffff
for (T, n, h) in [(32, 25, 42), (32, 25, 21), (32, 75, 42), (32, 5, 42), (4, 25, 42), (1, 25, 42)]:
    c, w = 32, h
    x = torch.randn(T, n, h, w, c, device='cuda')
    dz = torch.randn(T, n, h, w, c, device='cuda')
    ps = 9 * c * c + 64
    wt = torch.randn(T, ps, device='cuda') * 0.1
    dw = torch.empty(T, ps, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, c)
    scr = torch.empty(sb, dtype=torch.uint8, device='cuda')
    out, ref = [], None
    for name, mode in (('fp32 rows', 0), ('bf16 strips', (ALL << 8) | 1), ('bf16 units', ((ALL | (1 << 21)) << 8) | 1)):
        lib.mi_conv_set_split_bf16(mode)
        run = lambda: _lib.check(lib.mi_conv3x3_bwd(st(), vp(x), vp(dz), vp(wt), ps, T, n, h, w, c, c, 1, None, vp(dw), ps, vp(scr), sb))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ref = dw.clone() if mode == 0 else ref
        out.append(f'{name} {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (vs fp32 rows {float((dw - ref).norm() / ref.norm()):.1e})')
    print(f'T={T} n={n} {h}x{w}: ' + ' | '.join(out), flush=True)
lib.mi_conv_set_split_bf16(1)
