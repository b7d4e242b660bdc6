#!/usr/bin/env python3
"""Socket power, shader clock and energy per call of single kernels in a loop (the conv workloads run at the MI355X's 1400 W socket cap,
DESIGN.md 5 / 8c: at the cap a kernel's cost is its ENERGY).  Each primitive of the C ABI is launched back to back for ~2.5 s while a
read-only `rocm-smi -c -P` child process is sampled; reported per primitive and operand form: us per call, median shader clock, median
socket power, J per call (power x time), and the same with all-zero operands (how much of the power is data toggling).
Geometry: block 2 of cfg2 (32 tasks x 25 images, 42 x 42 x 32) unless --n / --hw / --c say otherwise."""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exploring_meta_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tasks', type=int, default=32)
ap.add_argument('--n', type=int, default=25)
ap.add_argument('--hw', type=int, default=42)
ap.add_argument('--c', type=int, default=32)
ap.add_argument('--seconds', type=float, default=2.5)
ap.add_argument('--only', default='', help='substring of the primitive names to run (randn operands, split form only): for ablation builds selected with MI_MAML_LIB')
args = ap.parse_args()
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
T, n, h, c = args.tasks, args.n, args.hw, args.c
ps = 9 * c * c + 64
sb = lib.mi_kernel_scratch_bytes(T, n, h, h, c)
scr = torch.empty(sb, dtype=torch.uint8, device='cuda')


def tensors(zero):
    mk = (lambda *s: torch.zeros(*s, device='cuda')) if zero else (lambda *s: torch.randn(*s, device='cuda'))
    return dict(x=mk(T, n, h, h, c), x1=mk(T, n, h, h, c), dz=mk(T, n, h, h, c), dz1=mk(T, n, h, h, c), w=mk(T, ps) * 0.1, w1=mk(T, ps) * 0.1,
                z=torch.empty(T, n, h, h, c, device='cuda'), zd=torch.empty(T, n, h, h, c, device='cuda'), dx=torch.empty(T, n, h, h, c, device='cuda'),
                dw=torch.empty(T, ps, device='cuda'), mu=torch.zeros(T, c, device='cuda'), rstd=torch.ones(T, c, device='cuda'),
                m1=torch.empty(T, c, device='cuda'), m2=torch.empty(T, c, device='cuda'))


def prims(d):
    return {
        'conv fwd + stats': lambda: lib.mi_conv3x3_bn_stats(st(), vp(d['x']), vp(d['w']), ps, T, n, h, h, c, c, 1, vp(d['z']), vp(d['mu']), vp(d['rstd']), vp(scr), sb),
        'dgrad + wgrad': lambda: lib.mi_conv3x3_bwd(st(), vp(d['x']), vp(d['dz']), vp(d['w']), ps, T, n, h, h, c, c, 1, vp(d['dx']), vp(d['dw']), ps, vp(scr), sb),
        'wgrad only': lambda: lib.mi_conv3x3_bwd(st(), vp(d['x']), vp(d['dz']), vp(d['w']), ps, T, n, h, h, c, c, 1, None, vp(d['dw']), ps, vp(scr), sb),
        'tangent conv (2 terms)': lambda: lib.mi_conv3x3_tangent(st(), vp(d['x']), vp(d['w']), vp(d['x1']), vp(d['w1']), ps, vp(d['z']), vp(d['mu']), vp(d['rstd']),
                                                                 T, n, h, h, c, c, 1, vp(d['zd']), vp(d['m1']), vp(d['m2']), vp(scr), sb),
        'tangent dgrad + wgrad (2 terms)': lambda: lib.mi_conv3x3_bwd2(st(), vp(d['x']), vp(d['dz']), vp(d['x1']), vp(d['dz1']), vp(d['w']), vp(d['w1']), ps,
                                                                         T, n, h, h, c, c, 1, vp(d['dx']), vp(d['dw']), ps, vp(scr), sb),
        'stream copy (HBM only)': lambda: lib.mi_stream_copy(st(), vp(d['x']), vp(d['z']), C.c_size_t(d['x'].numel() * 4)),
    }


def smi():
    txt = subprocess.run(['rocm-smi', '-c', '-P'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    m, w = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt), re.search(r'Power \(W\): ([0-9.]+)', txt)
    return (int(m.group(1)) if m else None, float(w.group(1)) if w else None)


def measure(run):
    for _ in range(5):
        _lib.check(run())
    torch.cuda.synchronize()
    t_end, samples, calls = time.perf_counter() + args.seconds, [], 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() < t_end:
        p = subprocess.Popen(['rocm-smi', '-c', '-P'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        while p.poll() is None:
            for _ in range(16):
                run()
            calls += 16
            torch.cuda.current_stream().synchronize()             # the host stays at most 16 calls ahead of the GPU
        txt = p.stdout.read()
        m, w = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt), re.search(r'Power \(W\): ([0-9.]+)', txt)
        if m and w:
            samples.append((int(m.group(1)), float(w.group(1))))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / calls
    samples = samples[len(samples) // 3:] or samples          # the first third still sees the previous state
    clk = sorted(s[0] for s in samples)[len(samples) // 2]
    pw = sorted(s[1] for s in samples)[len(samples) // 2]
    return us, clk, pw


time.sleep(1.0)
print(f'idle: sclk {smi()[0]} MHz, {smi()[1]} W', flush=True)
print(f'geometry: {T} tasks x {n} images, {h} x {h} x {c}', flush=True)
print('| primitive | operands | form | us / call | sclk MHz | socket W | J / call |', flush=True)
print('|---|---|---|---|---|---|---|', flush=True)
for zero in ((False,) if args.only else (False, True)):
    d = tensors(zero)
    for form, on in ((('split-bf16', 1),) if args.only else (('split-bf16', 1), ('fp32 pipe', 0))):
        lib.mi_conv_set_split_bf16(on)
        for name, run in prims(d).items():
            if ('stream copy' in name and not on) or (args.only and args.only not in name):
                continue
            us, clk, pw = measure(run)
            print(f'| {name} | {"zeros" if zero else "randn"} | {form if "copy" not in name else "-"} | {us:.1f} | {clk} | {pw:.0f} | {pw * us * 1e-6:.4f} |', flush=True)
lib.mi_conv_set_split_bf16(1)
