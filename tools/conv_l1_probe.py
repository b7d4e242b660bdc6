#!/usr/bin/env python3
"""One conv launch shape for counter passes (rocprofv3 --pmc ... -- python3 tools/conv_l1_probe.py [bf]): the block-2 forward conv of the
headline configuration (32 tasks x 25 images x 42 x 42 x 32 channels), five launches of the 1-term and five of the 2-term kernel."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd import _lib  # noqa: E402

lib = _lib.load()
lib.mi_conv_set_split_bf16(1 if 'bf' in sys.argv[1:] else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
T, n, h, w, c = 32, 25, 42, 42, 32
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
for terms in (1, 2):
    for _ in range(5):
        _lib.check(lib.mi_conv3x3_tangent(st(), vp(x0), vp(w0), vp(x1) if terms == 2 else None, vp(w1) if terms == 2 else None, w0.shape[1],
                                          vp(z), vp(mu), vp(rs), T, n, h, w, c, c, 1, vp(zd), vp(m1), vp(m2), vp(scr), sb))
torch.cuda.synchronize()
print('done')
