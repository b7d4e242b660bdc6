#!/usr/bin/env python3
"""Where a launch of the 16x16x32 stride-1 convolution (csrc/conv_b16.h) spends a wave's time: in-kernel shader-clock totals of waves 0 and 4
of workgroup (0, 0, 0) -- the two waves of SIMD 0 in an 8-wave workgroup -- per phase (weight staging, tile heads, K loops, epilogues), from
a DIAGNOSTIC build of the library (-DMI_B16_STAMPS; no stamp executes in the production build):

    cd exploring_meta_amd/csrc && hipcc $(CXXFLAGS) -DMI_B16_STAMPS -c conv_mfma.hip -o /tmp/conv_stamps.o && \
        hipcc -shared -fPIC --offload-arch=gfx950 /tmp/conv_stamps.o <the other objects> -o libmi_maml_stamps.so
    MI_MAML_LIB=exploring_meta_amd/csrc/libmi_maml_stamps.so python tools/conv_b16_stamps.py        (tools/r5_stamps.sh does both on the GPU box)"""
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
lib.mi_conv_set_split_bf16(1)
T, n, h, w, c = 32, 25, 42, 42, 32
x0 = torch.relu(torch.randn(T, n, h, w, c, device='cuda'))
x1 = torch.randn(T, n, h, w, c, device='cuda')
w0 = torch.randn(T, 9 * c * c + 64, device='cuda') * 0.1
w1 = torch.randn(T, 9 * c * c + 64, device='cuda') * 0.1
z = torch.randn(T, n, h, w, c, device='cuda')
mu, rs = torch.zeros(T, c, device='cuda'), torch.ones(T, c, device='cuda')
zd = torch.empty_like(z)
m1, m2 = torch.empty(T, c, device='cuda'), torch.empty(T, c, device='cuda')
sb = lib.mi_kernel_scratch_bytes(T, n, h, w, c)
scr = torch.empty(sb, dtype=torch.uint8, device='cuda')
buf = torch.zeros(16, dtype=torch.int64, device='cuda')
for b16 in (2, 0):
    lib.mi_conv_set_b16(b16)
    for terms in (2, 1):
        def run():
            _lib.check(lib.mi_conv3x3_tangent(st(), vp(x0), vp(w0), vp(x1) if terms == 2 else None, vp(w1) if terms == 2 else None, w0.shape[1],
                                              vp(z), vp(mu), vp(rs), T, n, h, w, c, c, 1, vp(zd), vp(m1), vp(m2), vp(scr), sb))
        for _ in range(200):                                    # the sustained regime: the clock the chip holds depends on the load
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        if b16 == 0:
            print(f'32x32x16 kernel, {terms} term(s): conv + finalize launches {us:.1f} us')
            continue
        buf.zero_()
        lib.mi_debug_conv_stamps(C.c_void_p(buf.data_ptr()))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        lib.mi_debug_conv_stamps(None)
        s = buf.cpu().numpy().astype(np.int64)
        print(f'16x16x32 kernel, {terms} term(s): conv + finalize launches {us:.1f} us (production timing of this diagnostic build: stamps cost cycles)')
        for wv in (0, 1):
            a = s[8 * wv:8 * wv + 8]
            if a[7] == 0:
                print(f'  wave {4 * wv}: no stamps (4-wave workgroup)' if wv else '  wave 0: no stamps -- is this the -DMI_B16_STAMPS build?')
                continue
            tiles = int(a[7])
            tot = int(a[6] - a[0])
            mfma = (2 if terms == 2 else 1) * 216 * 16
            print(f'  wave {4 * wv}: {tiles} tiles, {tot} cycles in all: weights staged after {int(a[1] - a[0])}; per tile: K loop {a[3] / tiles:.0f} '
                  f'(its MFMAs alone: {mfma} of pipe time), epilogue {a[4] / tiles:.0f} (of which the combination of the three accumulators {a[2] / tiles:.0f}); after the last tile (reduction, fold) {int(a[6] - a[5])}')
