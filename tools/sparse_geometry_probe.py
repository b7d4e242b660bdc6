#!/usr/bin/env python3
"""Is block 1's Gram-path weight gradient (input_gram + sparse_wgrad + gram_wgrad) of a task the same bits whether the task is launched alone or as
one of T (the launch cuts every task into nblk = f(T) shares)?  Same inputs replicated T times; task 0 of every launch compared with the T = 1 launch.
    python tools/sparse_geometry_probe.py            (MI_SPARSE_WGRAD_BF16=0/1 selects the form)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from exploring_meta_amd import _lib  # noqa: E402
from gpu_utils import ptr, stream  # noqa: E402
import test_gpu_tangent_kernels as TK  # noqa: E402


def run(lib, T, n=25, h=84, w=84, ci=3, co=32):
    hp, wp = h // 2, w // 2
    rep = lambda a: np.repeat(a[:1], T, axis=0)
    x = rep(TK._rand(90, (1, n, h, w, ci), 0.0, 255.0))
    w9, w9d = rep(TK._rand(91, (1, 9 * ci, co), -0.3 / 255, 0.3 / 255)), rep(TK._rand(92, (1, 9 * ci, co), -0.3 / 255, 0.3 / 255))
    gamma, beta = rep(TK._rand(93, (1, co), 0.1, 1.0)), rep(TK._rand(94, (1, co), -0.3, 0.3))
    gammad, betad = rep(TK._rand(95, (1, co))), rep(TK._rand(96, (1, co)))
    dp, dpd = rep(TK._rand(97, (1, n, hp, wp, co))), rep(TK._rand(98, (1, n, hp, wp, co)))
    pbuf, (og, ob, ow), pstride = TK._pack(T, [gamma, beta, w9])
    vbuf, (ogd, obd, owd), vstride = TK._pack(T, [gammad, betad, w9d], pad=7)
    xd, dpd_, dpdd_ = TK.dev(x), TK.dev(dp), TK.dev(dpd)
    sb = lib.mi_block1_scratch_bytes(T, n, h, w, ci, co)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    f32 = lambda *s: torch.zeros(s, device='cuda')
    mu, rstd, m1, m2 = f32(T, co), f32(T, co), f32(T, co), f32(T, co)
    gstride = 2 * co + 9 * ci * co + 11
    gb, hb = f32(T, gstride), f32(T, gstride)
    p, zhm, pd2, zhdm2 = (f32(T, n, hp, wp, co) for _ in range(4))
    arg = torch.full((T, n, hp, wp, co), 255, dtype=torch.uint8, device='cuda')
    a = _lib.MiBlock1Args(x=xd.data_ptr(), w=pbuf.data_ptr() + 4 * ow, wd=vbuf.data_ptr() + 4 * owd, gamma=pbuf.data_ptr() + 4 * og, beta=pbuf.data_ptr() + 4 * ob,
                          pstride=pstride, gammad=vbuf.data_ptr() + 4 * ogd, betad=vbuf.data_ptr() + 4 * obd, vstride=vstride, mu=mu.data_ptr(), rstd=rstd.data_ptr(),
                          m1=m1.data_ptr(), m2=m2.data_ptr(), dgamma=gb.data_ptr(), dbeta=gb.data_ptr() + 4 * co, gstride=gstride, rdgamma=hb.data_ptr(),
                          rdbeta=hb.data_ptr() + 4 * co, hstride=gstride, dp=dpd_.data_ptr(), dpd=dpdd_.data_ptr(), arg_in=arg.data_ptr(), zh_in=zhm.data_ptr(),
                          tasks=T, n=n, h=h, w_=w, ci=ci, co=co)

    def b1(mode, p_out=None, zh_out=None, arg_out=None, out0=None, out1=None, ostride=0):
        _lib.check(lib.mi_block1_run(stream(), mode, C.byref(a), ptr(p_out), ptr(zh_out), ptr(arg_out), ptr(out0), ptr(out1), ostride, ptr(scratch), sb))
    b1(0, out0=mu, out1=rstd, ostride=co)
    b1(1, p_out=p, zh_out=zhm, arg_out=arg)
    b1(2, out0=gb, out1=gb[:, co:], ostride=gstride)
    b1(4, out0=m1, out1=m2, ostride=co)
    b1(8, p_out=pd2, zh_out=zhdm2)
    b1(6, out0=hb, out1=hb[:, co:], ostride=gstride)
    ng = 32
    gs = lib.mi_input_gram_scratch_bytes(T, n, h, ci)
    gscr = torch.empty(gs, dtype=torch.uint8, device='cuda')
    G = torch.empty(T, ng, ng, dtype=torch.float64, device='cuda')
    _lib.check(lib.mi_input_gram(stream(), ptr(xd), T, n, h, w, ci, ptr(gscr), gs, ptr(G)))
    dwg, rdwg = f32(T, 9 * ci * co), f32(T, 9 * ci * co)
    _lib.check(lib.mi_block1_wgrad_gram(stream(), C.byref(a), ptr(G), 0, ptr(dwg), 9 * ci * co, ptr(scratch), sb))
    _lib.check(lib.mi_block1_wgrad_gram(stream(), C.byref(a), ptr(G), 1, ptr(rdwg), 9 * ci * co, ptr(scratch), sb))
    torch.cuda.synchronize()
    assert torch.equal(dwg[0], dwg[-1]) and torch.equal(rdwg[0], rdwg[-1])       # the replicas of one launch agree
    return dwg[0].cpu().numpy(), rdwg[0].cpu().numpy(), gb[0, :2 * co].cpu().numpy()


def main():
    lib = _lib.load()
    ref = run(lib, 1)
    for T in (2, 3, 5, 8, 32):
        got = run(lib, T)
        out = []
        for name, r, g in zip(('dW', 'RdW', 'dgamma|dbeta'), ref, got):
            d = np.abs(r - g)
            out.append(f'{name}: {int((r != g).sum())} of {r.size} differ, max rel {float(d.max() / np.abs(r).max()):.2e}')
        print(f'T = {T} against T = 1 -- ' + '; '.join(out))


if __name__ == '__main__':
    main()
