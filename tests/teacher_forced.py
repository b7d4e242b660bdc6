"""Teacher-forced per-step parity of one task of a fused second-order call (test infrastructure; CPU only -- safe to run in
child processes beside a process that holds the GPU).

The engine's own per-step state of task t (mi_debug_set_trace: theta_k, g_k, the vector fed to every Hessian-vector product
and its result) is handed over as numpy arrays; the reference arithmetic (oracle/vision_ref.py, autograd) is evaluated AT that
state in fp64 and in fp32 (the reference's own precision), so every step is a single forward/backward or one Hessian-vector
product and the chaotic amplification of the K-step map does not enter."""
import os
import sys
from collections import OrderedDict

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _unflatten(flat, shapes):
    out, off = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


TAU = 1e-5          # a decision is a "near-tie" if its fp64 margin (|u| at the window's maximum, or the gap to the runner-up) is below this
EXPLAIN_G, EXPLAIN_H = 2e-5, 2e-4      # steps above these get the near-tie analysis
MAX_TRIALS = 8


def _windows(u, hp, wp):
    """[N,C,H,W] -> [N,C,hp,wp,4], position q = 2*dy + dx (the scan order of max_pool2d's first-maximum rule)."""
    n, c = u.shape[0], u.shape[1]
    return u[:, :, :2 * hp, :2 * wp].reshape(n, c, hp, 2, wp, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, hp, wp, 4)


class DecisionNet:
    """The reference forward (oracle/vision_ref.py::model_forward for MiniImagenetCNN) with ReLU + MaxPool2d of every block written
    as `sum over the window of u * sel`, sel = one-hot(first maximum of u) * [u > 0] held CONSTANT: the same value and the same
    first and second derivatives as F.relu + F.max_pool2d (both are piecewise linear), but the decisions are explicit tensors, so a
    decision whose fp64 margin is at rounding level can be flipped and the exact gradient under the flipped assignment evaluated."""

    def __init__(self, x, y, spec):
        import torch
        self.torch, self.x, self.y, self.spec = torch, x, y, spec

    def forward(self, p, sels=None, record=None):
        import torch.nn.functional as F
        torch = self.torch
        x, base = self.x, self.spec['base']
        for i in range(base['layers']):
            z = F.conv2d(x, p[f'base.{i}.conv.weight'], p[f'base.{i}.conv.bias'], stride=1, padding=1)
            u = F.batch_norm(z, None, None, p[f'base.{i}.normalize.weight'], p[f'base.{i}.normalize.bias'], training=True, momentum=0.1, eps=1e-5)
            hp, wp = u.shape[2] // 2, u.shape[3] // 2
            uw = _windows(u, hp, wp)
            if sels is None or sels[i] is None:
                ud = uw.detach()
                arg = ud.argmax(dim=4, keepdim=True)                       # first maximal index
                sel = torch.zeros_like(ud).scatter_(4, arg, 1.0) * (ud.gather(4, arg) > 0).to(ud.dtype)
            else:
                sel = sels[i]
            if record is not None:
                record.append((uw.detach(), sel.detach() if torch.is_tensor(sel) else sel))
            x = (uw * sel).sum(dim=4)
        return F.linear(x.reshape(-1, self.spec['fc_in']), p['linear.weight'], p['linear.bias'])

    def grad_hvp(self, theta_flat, shapes, v_flat=None, sels=None, record=None, keep_graph=False):
        import torch.nn.functional as F
        torch = self.torch
        p = OrderedDict((n, t.clone().requires_grad_(True)) for n, t in _unflatten(theta_flat, shapes).items())
        loss = F.cross_entropy(self.forward(p, sels, record), self.y)
        g = torch.autograd.grad(loss, list(p.values()), create_graph=(v_flat is not None) or keep_graph)
        gf = torch.cat([t.reshape(-1) for t in g])
        hv = None
        if v_flat is not None:
            hv = torch.cat([t.reshape(-1) for t in torch.autograd.grad((gf * v_flat).sum(), list(p.values()), retain_graph=keep_graph)])
        return (gf if keep_graph else gf.detach()), (hv.detach() if hv is not None else None), float(loss.detach())


def _candidates(record, tau):
    """Near-tie decisions of a recorded forward: (block, flat window index, kind, position a, position b, margin)."""
    out = []
    for i, (uw, sel) in enumerate(record):
        srt, idx = uw.sort(dim=4, descending=True, stable=True)
        top, second = srt[..., 0], srt[..., 1]
        relu = (top.abs() < tau).nonzero()
        for row in relu:
            n, c, y, x = (int(t) for t in row)
            out.append(dict(block=i, at=(n, c, y, x), kind='relu', a=int(idx[n, c, y, x, 0]), b=-1, margin=float(top[n, c, y, x])))
        am = ((top > 0) & ((top - second) < tau)).nonzero()
        for row in am:
            n, c, y, x = (int(t) for t in row)
            out.append(dict(block=i, at=(n, c, y, x), kind='argmax', a=int(idx[n, c, y, x, 0]), b=int(idx[n, c, y, x, 1]),
                            margin=float(top[n, c, y, x] - second[n, c, y, x])))
    return out


def _apply(sels, cand):
    n, c, y, x = cand['at']
    s = sels[cand['block']]
    if cand['kind'] == 'relu':
        s[n, c, y, x, cand['a']] = 1.0 - s[n, c, y, x, cand['a']]
    else:
        va, vb = float(s[n, c, y, x, cand['a']]), float(s[n, c, y, x, cand['b']])
        s[n, c, y, x, cand['a']], s[n, c, y, x, cand['b']] = vb, va


def explain_step(net, theta_k, shapes, g_e, v, hv_e, g0, hv0):
    """The engine's (g_e, hv_e) against the fp64 arithmetic under the best decision assignment that differs from the fp64 one only at
    near-ties (margin < TAU).  Ranking of the candidates: first-order effect of every selection weight on <g_e - g0, g(sel)> (one
    double backward for all of them); then up to MAX_TRIALS exact re-evaluations, each flip kept if it lowers the combined residual.
    -> (err_g, err_h, flips kept)."""
    torch = net.torch
    rec = []
    net.forward(_unflatten(theta_k, shapes), None, rec)
    cands = _candidates(rec, TAU)
    sels = [sel.clone() for _, sel in rec]
    if not cands:
        return rel_err(g_e.numpy(), g0.numpy()), (rel_err(hv_e.numpy(), hv0.numpy()) if hv_e is not None else 0.0), []
    # scores: d<r, g(sel)>/d sel, contracted with every candidate's change of sel
    sreq = [s_.clone().requires_grad_(True) for s_ in sels]
    gf, _, _ = net.grad_hvp(theta_k, shapes, None, sreq, None, keep_graph=True)
    r = (g_e - g0)
    ds = torch.autograd.grad((gf * r).sum(), sreq, allow_unused=True)
    for cd in cands:
        n, c, y, x = cd['at']
        d = ds[cd['block']]
        if d is None:
            cd['score'] = 0.0
        elif cd['kind'] == 'relu':
            cd['score'] = float(d[n, c, y, x, cd['a']]) * (1.0 - 2.0 * float(sels[cd['block']][n, c, y, x, cd['a']]))
        else:
            cd['score'] = float(d[n, c, y, x, cd['b']] - d[n, c, y, x, cd['a']])
    cands.sort(key=lambda cd: -cd['score'])

    def objective(g, hv):
        eg = rel_err(g_e.numpy(), g.numpy())
        eh = rel_err(hv_e.numpy(), hv.numpy()) if hv_e is not None else 0.0
        return (eg / 1e-5) ** 2 + (eh / 1e-4) ** 2, eg, eh

    best, eg, eh = objective(g0, hv0)
    kept = []
    for cd in cands[:MAX_TRIALS]:
        if cd['score'] <= 0.0 and kept:
            break
        _apply(sels, cd)
        g1, hv1, _ = net.grad_hvp(theta_k, shapes, v, sels)
        j1, eg1, eh1 = objective(g1, hv1)
        if j1 < best:
            best, eg, eh = j1, eg1, eh1
            kept.append(dict(block=cd['block'], at=cd['at'], kind=cd['kind'], margin=cd['margin']))
        else:
            _apply(sels, cd)          # undo
    return eg, eh, kept


def teacher_forced_task(job):
    """job = dict(t, theta [K+1,P], g [K,P], lam_in [K,P], hv [K,P], data [2SW,C,H,W], labels [2SW], shots, ways, threads).
    -> dict(t, g64, g32, h64, h32 (lists over k: raw relative errors against the two legs), q64, q32 = (loss, accuracy,
    query-gradient error), gx, hx (lists over k: errors against the fp64 arithmetic under the near-tie-adjusted decisions), qx,
    flips (list over k + the query pass of the decisions that were flipped))."""
    import torch
    import torch.nn.functional as F
    from oracle import vision_ref as R
    torch.set_num_threads(int(job.get('threads', 4)))
    ways, shots = job['ways'], job['shots']
    spec = R.mini_imagenet_spec(ways)
    shapes = R.param_shapes(spec)
    theta, g_e, lam_in, hv_e = (torch.from_numpy(np.asarray(job[k])) for k in ('theta', 'g', 'lam_in', 'hv'))
    K = g_e.shape[0]
    xs64, ys, xq64, yq = R.prepare_batch(torch.from_numpy(job['data']).double(), torch.from_numpy(job['labels']), shots, ways)
    out = dict(t=job['t'], g64=[], g32=[], h64=[], h32=[], gx=[], hx=[], flips=[])
    explain = bool(job.get('explain', True))
    net = DecisionNet(xs64, ys, spec)
    for k in range(K):
        keep = {}
        for dt, tag in ((torch.float64, '64'), (torch.float32, '32')):
            p = OrderedDict((n, v.to(dt).clone().requires_grad_(True)) for n, v in _unflatten(theta[k], shapes).items())
            loss = F.cross_entropy(R.model_forward(xs64.to(dt), p, spec), ys)
            g = torch.autograd.grad(loss, list(p.values()), create_graph=True)
            v = _unflatten(lam_in[k].to(dt), shapes)
            hv = torch.autograd.grad(sum((gi * v[n]).sum() for gi, n in zip(g, p)), list(p.values()))
            gf, hf = torch.cat([x.detach().reshape(-1) for x in g]).double(), torch.cat([x.reshape(-1) for x in hv]).double()
            out['g' + tag].append(rel_err(g_e[k].numpy(), gf.numpy()))
            out['h' + tag].append(rel_err(hv_e[k].numpy(), hf.numpy()))
            keep[tag] = (gf, hf)
        gx, hx, flips = out['g64'][-1], out['h64'][-1], []
        if explain and (gx > EXPLAIN_G or hx > EXPLAIN_H):
            gx, hx, flips = explain_step(net, theta[k].double(), shapes, g_e[k].double(), lam_in[k].double(), hv_e[k].double(), *keep['64'])
        out['gx'].append(gx)
        out['hx'].append(hx)
        out['flips'].append(flips)
    for dt, tag in ((torch.float64, '64'), (torch.float32, '32')):
        pK = OrderedDict((n, v.to(dt).clone().requires_grad_(True)) for n, v in _unflatten(theta[K], shapes).items())
        logits = R.model_forward(xq64.to(dt), pK, spec)
        lq = F.cross_entropy(logits, yq)
        gq = torch.cat([x.reshape(-1) for x in torch.autograd.grad(lq, list(pK.values()))]).double()
        out['q' + tag] = (float(lq.detach()), float(R.accuracy(logits, yq)), rel_err(lam_in[K - 1].numpy(), gq.numpy()))
        if tag == '64':
            gq64 = gq
    out['qx'], qflips = out['q64'][2], []
    if explain and out['qx'] > EXPLAIN_G:
        qnet = DecisionNet(xq64, yq, spec)
        out['qx'], _, qflips = explain_step(qnet, theta[K].double(), shapes, lam_in[K - 1].double(), None, None, gq64, None)
    out['flips'].append(qflips)
    return out


def teacher_forced_all(trace, data, labels, shots, ways, tasks, workers=None, threads=4, timeout=900, explain=True):
    """Run `teacher_forced_task` for every task in `tasks` in CPU worker processes (plain `python teacher_forced.py jobs.npz
    out.json` children: the parent keeps the GPU, the workers never touch it).  trace: dict of [*, T, P] tensors from
    MetaEngine.set_trace."""
    import json
    import subprocess
    import tempfile
    tr = {k: trace[k].detach().cpu().numpy() for k in ('theta', 'g', 'lam_in', 'hv')}
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    tasks = list(tasks)
    if workers is None:
        workers = max(1, min(len(tasks), cores // threads, 4))       # + the parent: within the GPU box's 6-process guard (importing torch opens the device)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            mine = tasks[w::workers]
            if not mine:
                continue
            job, out = os.path.join(tmp, f'job{w}.npz'), os.path.join(tmp, f'out{w}.json')
            np.savez(job, tasks=np.array(mine), shots=shots, ways=ways, threads=threads, explain=int(explain), data=data[mine], labels=labels[mine],
                     **{k: v[:, mine] for k, v in tr.items()})
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), job, out], env=env), out))
        res = []
        for p, out in procs:
            try:
                rc = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q, _ in procs:
                    q.kill()
                raise RuntimeError('teacher-forced oracle worker timed out')
            if rc != 0:
                raise RuntimeError(f'teacher-forced oracle worker failed (exit {rc})')
            with open(out) as f:
                res += json.load(f)
    return sorted(res, key=lambda r: r['t'])


def _worker_main(job_path, out_path):
    import json
    j = np.load(job_path)
    res = []
    for i, t in enumerate(j['tasks']):
        res.append(teacher_forced_task(dict(t=int(t), theta=j['theta'][:, i], g=j['g'][:, i], lam_in=j['lam_in'][:, i], hv=j['hv'][:, i],
                                            data=j['data'][i], labels=j['labels'][i], shots=int(j['shots']), ways=int(j['ways']),
                                            threads=int(j['threads']), explain=bool(int(j['explain'])))))
    with open(out_path, 'w') as f:
        json.dump(res, f)


if __name__ == '__main__':
    _worker_main(sys.argv[1], sys.argv[2])
