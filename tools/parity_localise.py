#!/usr/bin/env python3
"""Localise a teacher-forced parity outlier of BASELINE config 2: for one task and one inner step,
where does the engine's support-set gradient leave the reference arithmetic?

At the engine's own theta_k (mi_debug_set_trace) this tool compares, against oracle/vision_ref.py in fp64 and fp32:
  * the step gradient g_k per parameter tensor (which layer carries the error);
  * every ConvBlock's output (mi_learner_forward with rep_layer = 1..4): relative error and the elements that differ by more than
    1e-4 of the tensor's scale (a flipped max-pool / ReLU decision shows as a handful of O(1) differences, a kernel error as a
    dense small one);
  * the backward alone (mi_learner_backward with the ORACLE's dlogits as cotangent);
  * block 1's stored pooling argmax byte (mi_block1_run, MI_B1_FWD) against the indices of torch's MaxPool2d after BN + ReLU
    (reference core_functions/vision_models.py:188-193), with the near-tie structure of every window that differs.
Test infrastructure: runs on the GPU box, never part of the timed path.  Output: a markdown report (stdout and --out)."""
import argparse
import ctypes as C
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from exploring_meta_amd import _lib  # noqa: E402
from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402
from oracle import vision_ref as R  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def unflatten(flat, shapes):
    out, off = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def block1_argmax(lib, theta_k, xs_nhwc, shapes):
    """Engine block-1 forward (statistics + lean forward kernel) on one task's support images -> (p, zh_at, arg byte)."""
    n, h, w, ci = xs_nhwc.shape
    co = shapes['base.0.conv.weight'][0]
    p = unflatten(theta_k, shapes)
    w9 = p['base.0.conv.weight'].permute(2, 3, 1, 0).reshape(9 * ci, co).contiguous()
    pbuf = torch.cat([p['base.0.normalize.weight'].reshape(-1), p['base.0.normalize.bias'].reshape(-1), w9.reshape(-1),
                      torch.zeros(5)]).float().cuda().contiguous()
    og, ob, ow = 0, co, 2 * co
    x = xs_nhwc.float().cuda().contiguous()
    sb = lib.mi_block1_scratch_bytes(1, n, h, w, ci, co)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    mu, rstd = torch.empty(co, device='cuda'), torch.empty(co, device='cuda')
    hp, wp = h // 2, w // 2
    pout, zh = torch.empty(n, hp, wp, co, device='cuda'), torch.empty(n, hp, wp, co, device='cuda')
    arg = torch.full((n, hp, wp, co), 255, dtype=torch.uint8, device='cuda')
    a = _lib.MiBlock1Args(x=x.data_ptr(), w=pbuf.data_ptr() + 4 * ow, gamma=pbuf.data_ptr() + 4 * og, beta=pbuf.data_ptr() + 4 * ob,
                          pstride=pbuf.numel(), mu=mu.data_ptr(), rstd=rstd.data_ptr(), tasks=1, n=n, h=h, w_=w, ci=ci, co=co)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    _lib.check(lib.mi_block1_run(st, 0, C.byref(a), None, None, None, vp(mu), vp(rstd), co, vp(scratch), sb))
    _lib.check(lib.mi_block1_run(st, 1, C.byref(a), vp(pout), vp(zh), vp(arg), None, None, 0, vp(scratch), sb))
    torch.cuda.synchronize()
    return pout.cpu(), zh.cpu(), arg.cpu(), mu.cpu(), rstd.cpu()


def torch_block1(xs_nchw, p, dt):
    """The reference's block 1 in dtype dt: -> u (BN output), pooled p, argmax position 0..3 per window (4 where ReLU is off)."""
    x = xs_nchw.to(dt)
    z = F.conv2d(x, p['base.0.conv.weight'].to(dt), p['base.0.conv.bias'].to(dt), stride=1, padding=1)
    u = F.batch_norm(z, None, None, p['base.0.normalize.weight'].to(dt), p['base.0.normalize.bias'].to(dt), training=True, eps=1e-5)
    a = F.relu(u)
    pooled, idx = F.max_pool2d(a, 2, 2, return_indices=True)
    H, W = u.shape[2], u.shape[3]
    iy, ix = idx // W, idx % W
    pos = (iy % 2) * 2 + (ix % 2)
    pos = torch.where(pooled > 0, pos, torch.full_like(pos, 4))
    return z, u, pooled, pos        # NCHW


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tasks', default='7')
    ap.add_argument('--step', type=int, default=0)
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ways, shots, K, lr, T = 5, 5, 5, 0.5, 32
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    shapes = R.param_shapes(spec)
    th0 = OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic.ref_init_weights(shapes, 42).items())
    theta = R.flatten_params(th0).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    eng = MetaEngine(mspec)
    lib = _lib.load()
    trace = eng.set_trace(T, K)
    eng.meta_batch(theta, torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), shots, K, lr)
    torch.cuda.synchronize()
    lines = []

    def say(s=''):
        print(s, flush=True)
        lines.append(s)

    k = args.step
    for t in [int(x) for x in args.tasks.split(',')]:
        say(f'## cfg2 task {t}, inner step {k}')
        th_k = trace['theta'][k, t].cpu()
        g_eng = trace['g'][k, t].cpu()
        xs, ys, _, _ = R.prepare_batch(torch.from_numpy(data[t]).double(), torch.from_numpy(labels[t]), shots, ways)
        legs = {}
        for dt, tag in ((torch.float64, 'fp64'), (torch.float32, 'fp32')):
            p = OrderedDict((n, v.to(dt).clone().requires_grad_(True)) for n, v in unflatten(th_k, shapes).items())
            feats, x = [], xs.to(dt)
            for i in range(4):
                x = R.conv_block(x, p, i, spec['base'])
                feats.append(x)
            logits = F.linear(x.reshape(-1, spec['fc_in']), p['linear.weight'], p['linear.bias'])
            loss = F.cross_entropy(logits, ys)
            g = torch.autograd.grad(loss, list(p.values()), retain_graph=True)
            dlog = torch.autograd.grad(loss, logits, retain_graph=True)[0].detach()
            legs[tag] = dict(p=p, feats=[f.detach() for f in feats], logits=logits.detach(), g=OrderedDict(zip(p.keys(), g)), dlog=dlog,
                             loss=float(loss))
        say(f'loss fp64 {legs["fp64"]["loss"]:.9f}  fp32 {legs["fp32"]["loss"]:.9f}')
        say()
        say('| parameter | |g| fp64 | engine vs fp64 | engine vs fp32 | fp32 vs fp64 |')
        say('|---|---|---|---|---|')
        ge = unflatten(g_eng, shapes)
        for n in shapes:
            a64, a32 = legs['fp64']['g'][n].numpy(), legs['fp32']['g'][n].numpy()
            if np.linalg.norm(a64) < 1e-12:
                continue
            say(f'| {n} | {np.linalg.norm(a64):.3e} | {rel(ge[n].numpy(), a64):.2e} | {rel(ge[n].numpy(), a32):.2e} | {rel(a32, a64):.2e} |')
        g64 = torch.cat([v.reshape(-1) for v in legs['fp64']['g'].values()]).numpy()
        g32 = torch.cat([v.reshape(-1) for v in legs['fp32']['g'].values()]).numpy()
        say(f'| ALL | {np.linalg.norm(g64):.3e} | {rel(g_eng.numpy(), g64):.2e} | {rel(g_eng.numpy(), g32):.2e} | {rel(g32, g64):.2e} |')
        say()
        # ---- forward, block by block
        xs_dev = xs.float().cuda().contiguous()[None]
        thk_dev = th_k.float().cuda().contiguous()[None]
        say('| block output | engine vs fp64 | engine vs fp32 | fp32 vs fp64 | elements off by > 1e-4 max|p| (engine-fp64 / engine-fp32 / fp32-fp64) |')
        say('|---|---|---|---|---|')
        for L in range(1, 5):
            _, rep = eng.learner_forward(thk_dev, xs_dev, rep_layer=L, want_logits=False)
            torch.cuda.synchronize()
            e = rep[0].double().cpu().numpy()
            f64, f32 = legs['fp64']['feats'][L - 1].numpy(), legs['fp32']['feats'][L - 1].double().numpy()
            thr = 1e-4 * np.abs(f64).max()
            say(f'| {L} | {rel(e, f64):.2e} | {rel(e, f32):.2e} | {rel(f32, f64):.2e} | {(np.abs(e - f64) > thr).sum()} / '
                f'{(np.abs(e - f32) > thr).sum()} / {(np.abs(f32 - f64) > thr).sum()} of {e.size} |')
        lg, _ = eng.learner_forward(thk_dev, xs_dev)
        torch.cuda.synchronize()
        say(f'| logits | {rel(lg[0].cpu().numpy(), legs["fp64"]["logits"].numpy()):.2e} | {rel(lg[0].cpu().numpy(), legs["fp32"]["logits"].numpy()):.2e} | '
            f'{rel(legs["fp32"]["logits"].numpy(), legs["fp64"]["logits"].numpy()):.2e} | |')
        say()
        # ---- backward alone: the fp64 leg's dlogits as cotangent
        cot = legs['fp64']['dlog']
        gb = eng.learner_backward(thk_dev, xs_dev, cot.float().cuda()[None])
        torch.cuda.synchronize()
        gb = unflatten(gb[0].cpu(), shapes)
        say('backward alone (cotangent = the fp64 leg\'s dlogits), per parameter vs the fp64 / fp32 legs\' own gradients:')
        say()
        say('| parameter | engine vs fp64 | engine vs fp32 |')
        say('|---|---|---|')
        for n in shapes:
            a64, a32 = legs['fp64']['g'][n].numpy(), legs['fp32']['g'][n].numpy()
            if np.linalg.norm(a64) < 1e-12:
                continue
            say(f'| {n} | {rel(gb[n].numpy(), a64):.2e} | {rel(gb[n].numpy(), a32):.2e} |')
        say()
        # ---- block-1 argmax byte
        xs_nhwc = xs.permute(0, 2, 3, 1).contiguous()
        p_e, zh_e, arg_e, mu_e, rstd_e = block1_argmax(lib, th_k, xs_nhwc, shapes)
        pk = unflatten(th_k, shapes)
        res = {}
        for dt, tag in ((torch.float64, 'fp64'), (torch.float32, 'fp32')):
            z, u, pooled, pos = torch_block1(xs, pk, dt)
            res[tag] = dict(z=z, u=u, pooled=pooled, pos=pos.permute(0, 2, 3, 1).contiguous())
        ae = arg_e.long()
        n_win = ae.numel()
        d64, d32 = (ae != res['fp64']['pos']), (ae != res['fp32']['pos'])
        dl = res['fp64']['pos'] != res['fp32']['pos']
        say(f'block-1 argmax byte over {n_win} (window, channel) pairs: engine != fp64 leg {int(d64.sum())}, engine != fp32 leg {int(d32.sum())}, '
            f'fp32 leg != fp64 leg {int(dl.sum())}, engine differs from BOTH {int((d64 & d32).sum())}')
        # tie structure in the legs: windows whose maximum is attained more than once
        for tag in ('fp64', 'fp32'):
            a = F.relu(res[tag]['u'])
            nb, c, H, W = a.shape
            aw = a.reshape(nb, c, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(nb, H // 2, W // 2, c, 4)
            mx = aw.max(dim=4, keepdim=True).values
            ties = ((aw == mx).sum(dim=4) > 1) & (mx[..., 0] > 0)
            say(f'  {tag} leg: windows with an exactly tied positive maximum: {int(ties.sum())}')
            res[tag]['ties'] = ties
            res[tag]['aw'] = aw
        both = (d64 & d32).nonzero()
        say(f'  first windows where the engine differs from both legs (image, wy, wx, channel): engine arg / fp64 / fp32, u of the 4 positions (fp64), tied in fp32 leg?')
        for row in both[:20]:
            i, wy, wx, c = [int(v) for v in row]
            say(f'    ({i},{wy},{wx},{c}): {int(ae[i, wy, wx, c])} / {int(res["fp64"]["pos"][i, wy, wx, c])} / {int(res["fp32"]["pos"][i, wy, wx, c])}  u64 = '
                f'{[float(v) for v in res["fp64"]["aw"][i, wy, wx, c]]}  u32 = {[float(v) for v in res["fp32"]["aw"][i, wy, wx, c]]}  '
                f'tied32 {bool(res["fp32"]["ties"][i, wy, wx, c])} tied64 {bool(res["fp64"]["ties"][i, wy, wx, c])}')
        # the engine's pooled value against the legs
        pe = p_e.permute(0, 3, 1, 2).double().numpy()
        say(f'  pooled value p1: engine vs fp64 {rel(pe, res["fp64"]["pooled"].numpy()):.2e}, vs fp32 {rel(pe, res["fp32"]["pooled"].double().numpy()):.2e}')
        say()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
