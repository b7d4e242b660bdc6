#!/usr/bin/env python3
"""CPU estimate behind DESIGN.md 8c lead 5: error of a K = 288 dot product (one 3x3x32 conv output) against fp64 for
  (a) an fp32 multiply-add chain, (b) the shipped three-plane bf16 split with six products, (c) a two-plane fp16 split with three
  products (hh, hm, mh) and (d) the same plus mm, operands scaled by a per-tensor power of two so that the largest magnitude sits at 2^14.
Products of 16-bit pieces are exact in fp32; every accumulation is rounded to fp32 (the MFMA accumulates K = 16 at a time: modelled as an
exact 16-term sum rounded once, which is optimistic by less than one rounding per step for all four alike)."""
import numpy as np

rng = np.random.default_rng(0)
N, K = 20000, 288


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16(x):
    h = bf16(x); m = bf16((x - h).astype(np.float32)); l = bf16((x - h - m).astype(np.float32))
    return h, m, l


def split_fp16(x, with_scale=True):
    s = np.float32(2.0 ** (14 - np.ceil(np.log2(np.abs(x).max())))) if with_scale else np.float32(1)
    xs = (x * s).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    m = (xs - h).astype(np.float16).astype(np.float32)
    return h, m, s


def acc16(prods):                       # prods [N, K] float64 exact products -> fp32 accumulation in K = 16 steps
    acc = np.zeros(prods.shape[0], np.float32)
    for k in range(0, prods.shape[1], 16):
        acc = (acc.astype(np.float64) + prods[:, k:k + 16].sum(axis=1)).astype(np.float32)
    return acc


for sx, sw, label in ((1.0, 0.1, 'activations ~1, weights ~0.1'), (1e-6, 0.1, 'cotangents ~1e-6, weights ~0.1')):
    x = (rng.standard_normal((N, K)) * sx).astype(np.float32)
    x *= (rng.random((N, K)) > 0.5)                      # half the activations are ReLU zeros
    w = (rng.standard_normal((N, K)) * sw).astype(np.float32)
    ref = (x.astype(np.float64) * w.astype(np.float64)).sum(axis=1)
    scale = np.sqrt((x.astype(np.float64) ** 2 * w.astype(np.float64) ** 2).sum(axis=1))       # natural size of the sum
    a = np.zeros(N, np.float32)
    for k in range(K):
        a = np.float32(a + x[:, k] * w[:, k]) if False else (a.astype(np.float64) + (x[:, k].astype(np.float64) * w[:, k].astype(np.float64)).astype(np.float32)).astype(np.float32)
    xh, xm, xl = split_bf16(x); wh, wm, wl = split_bf16(w)
    d = lambda p, q: p.astype(np.float64) * q.astype(np.float64)
    b = acc16(d(xl, wh) + d(xh, wl) + d(xm, wm) + d(xm, wh) + d(xh, wm) + d(xh, wh))
    fh, fm, fs = split_fp16(x); gh, gm, gs = split_fp16(w)
    c3 = acc16(d(fh, gm) + d(fm, gh) + d(fh, gh)) / np.float32(fs * gs)
    c4 = acc16(d(fm, gm) + d(fh, gm) + d(fm, gh) + d(fh, gh)) / np.float32(fs * gs)
    rms = lambda v: float(np.sqrt(np.mean(((v.astype(np.float64) - ref) / scale) ** 2)))
    print(f'{label}: rms error / |terms|_2:  fp32 chain {rms(a):.2e} | bf16 x3, 6 products {rms(b):.2e} | fp16 x2, 3 products {rms(c3):.2e} | fp16 x2, 4 products {rms(c4):.2e}')
