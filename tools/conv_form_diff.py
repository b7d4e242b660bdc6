#!/usr/bin/env python3
"""Elementwise difference of the two operand forms (split bf16 / fp32 pipe, mi_conv_set_split_bf16) of the stride-1 hidden convolutions
on the same inputs, through the kernel-level entries: forward + statistics, dgrad, 2-term tangent forward.  Expected: ~5e-7 of the
tensor's scale everywhere (fp32 rounding of a 288-term dot product), no element above 1e-5 of it.      python tools/conv_form_diff.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
torch.manual_seed(0)
for (T, n, h, w) in [(1, 5, 42, 42), (2, 5, 21, 21), (3, 5, 10, 10), (1, 10, 42, 42), (4, 3, 42, 42)]:
    c = 32
    x0 = torch.relu(torch.randn(T, n, h, w, c, device='cuda')) * 3
    x1 = torch.randn(T, n, h, w, c, device='cuda')
    ps = 9 * c * c + 64
    w0 = torch.randn(T, ps, device='cuda') * 0.1
    w1 = torch.randn(T, ps, device='cuda') * 0.1
    dz = torch.randn(T, n, h, w, c, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, c)
    scr = torch.empty(sb, dtype=torch.uint8, device='cuda')
    res = {}
    for bf in (0, 1):
        lib.mi_conv_set_split_bf16(bf)
        z = torch.full((T, n, h, w, c), float('nan'), device='cuda')
        mu, rs = torch.empty(T, c, device='cuda'), torch.empty(T, c, device='cuda')
        _lib.check(lib.mi_conv3x3_bn_stats(st(), vp(x0), vp(w0), ps, T, n, h, w, c, c, 1, vp(z), vp(mu), vp(rs), vp(scr), sb))
        dx = torch.full((T, n, h, w, c), float('nan'), device='cuda')
        dw = torch.full((T, ps), float('nan'), device='cuda')
        _lib.check(lib.mi_conv3x3_bwd(st(), vp(x0), vp(dz), vp(w0), ps, T, n, h, w, c, c, 1, vp(dx), vp(dw), ps, vp(scr), sb))
        zd = torch.full((T, n, h, w, c), float('nan'), device='cuda')
        m1, m2 = torch.empty(T, c, device='cuda'), torch.empty(T, c, device='cuda')
        _lib.check(lib.mi_conv3x3_tangent(st(), vp(x0), vp(w1), vp(x1), vp(w0), ps, vp(z), vp(mu), vp(rs), T, n, h, w, c, c, 1, vp(zd), vp(m1), vp(m2), vp(scr), sb))
        torch.cuda.synchronize()
        res[bf] = dict(z=z.clone(), mu=mu.clone(), rs=rs.clone(), dx=dx.clone(), zd=zd.clone(), m1=m1.clone(), m2=m2.clone())
    # fp64 reference of z through torch
    xr = x0.double().permute(0, 4, 1, 2, 3)  # T c n h w
    for k in ('z', 'dx', 'zd', 'mu', 'rs', 'm1', 'm2'):
        a, b = res[0][k].double(), res[1][k].double()
        d = (a - b).abs()
        i = int(d.argmax())
        idx = np.unravel_index(i, d.shape)
        print(f'T={T} n={n} {h}x{w} {k}: max|bf-fp32| {float(d.max()):.3e} (scale {float(a.abs().max()):.3e}) at {tuple(int(v) for v in idx)}  n_bad(>1e-5*scale) {int((d > 1e-5 * a.abs().max()).sum())}', flush=True)
