#!/usr/bin/env python3
"""Where a stride-1 conv launch spends its time at few tasks per call: in-kernel stamps of workgroup 0 (weights staged, first tile, all
tiles, epilogue) against the launch's duration (HIP events), through the kernel-level entry mi_conv3x3_tangent (2-term) / mi_conv3x3_bn_stats."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd import _lib  # noqa: E402

lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
BF_MODES = (0, 1, 2) if '--bf' in sys.argv else (None,)      # operand forms: fp32 pipe, three bf16 planes, two fp16 planes
for T in (1, 4, 32):
    n, h, w, c = 25, 42, 42, 32
    x0 = torch.randn(T, n, h, w, c, device='cuda')
    x1 = torch.randn(T, n, h, w, c, device='cuda')
    w0 = torch.randn(T, 9 * c * c + 64, device='cuda') * 0.1
    w1 = torch.randn(T, 9 * c * c + 64, device='cuda') * 0.1
    z = torch.randn(T, n, h, w, c, device='cuda')
    mu, rs = torch.zeros(T, c, device='cuda'), torch.ones(T, c, device='cuda')
    zd = torch.empty_like(z)
    m1, m2 = torch.empty(T, c, device='cuda'), torch.empty(T, c, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, c)
    scr = torch.empty(sb, dtype=torch.uint8, device='cuda')
    buf = torch.zeros(8, dtype=torch.int64, device='cuda')
    for terms, bf in [(t, b) for t in (1, 2) for b in BF_MODES]:
        if bf is not None:
            lib.mi_conv_set_split_bf16(bf)

        def run():
            _lib.check(lib.mi_conv3x3_tangent(st(), vp(x0), vp(w0), vp(x1) if terms == 2 else None, vp(w1) if terms == 2 else None, w0.shape[1],
                                              vp(z), vp(mu), vp(rs), T, n, h, w, c, c, 1, vp(zd), vp(m1), vp(m2), vp(scr), sb))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        lib.mi_debug_conv_stamps(C.c_void_p(buf.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.mi_debug_conv_stamps(None)
        s = buf.cpu().numpy().astype(np.int64)
        d = [int(s[i + 1] - s[i]) for i in range(4)]
        print(f'T={T} terms={terms} operand form {bf}: conv + finalize (form 2: + two largest-magnitude reductions) launches {e0.elapsed_time(e1) / 10 * 1e3:.1f} us | wg0 cycles: weights {d[0]}, first tile {d[1]}, '
              f'remaining tiles {d[2]}, epilogue {d[3]}, total {int(s[4] - s[0])}', flush=True)
