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


TAU = 3e-6          # a decision is a "near-tie" if its fp64 margin (|u| at the window's maximum, or the gap to the runner-up) is below this
                    # (the largest margin any run has flipped: 7.4e-7 on cfg2 / cfg3, 1.4e-6 on cfg4 at 256 tasks)
EXPLAIN_G, EXPLAIN_H = 1e-5, 1e-4      # steps above these get the near-tie analysis (the search stops below half of them)
MAX_TRIALS = 28


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

    def trunk(self, x, p, base, prefix, sels=None, record=None):
        import torch.nn.functional as F
        torch = self.torch
        for i in range(base['layers']):
            z = F.conv2d(x, p[f'{prefix}{i}.conv.weight'], p[f'{prefix}{i}.conv.bias'], stride=1, padding=1)
            u = F.batch_norm(z, None, None, p[f'{prefix}{i}.normalize.weight'], p[f'{prefix}{i}.normalize.bias'], training=True, momentum=0.1, eps=1e-5)
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
        return x

    def forward(self, p, sels=None, record=None):
        import torch.nn.functional as F
        x = self.trunk(self.x, p, self.spec['base'], 'base.', sels, record)
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


def explain(evaluate, rec, g_e, hv_e, g0, hv0):
    """The engine's (g_e, hv_e) against the fp64 arithmetic under the best decision assignment that differs from the fp64 one only at
    near-ties (margin < TAU).  `evaluate(sels, want_hv, graph)` -> (g, hv): the exact fp64 gradient (and Hessian-vector product)
    under the decision tensors `sels` (graph=True: g stays attached to `sels` for the ranking); `rec`: the (u windows, sel) record
    of the fp64 forward.  Search: candidates ranked by the first-order effect of their selection weights on <residual, g(sel)>
    (one double backward ranks all of them), then by ascending margin; each trial is an exact re-evaluation, a flip is kept if it
    lowers the combined residual, and the ranking is redone after every kept flip.  -> (err_g, err_h, flips kept)."""
    import torch
    cands = _candidates(rec, TAU)
    sels = [sel.clone() for _, sel in rec]

    def objective(g, hv):
        eg = rel_err(g_e.numpy(), g.numpy())
        eh = rel_err(hv_e.numpy(), hv.numpy()) if hv_e is not None else 0.0
        return (eg / 1e-5) ** 2 + (eh / 1e-4) ** 2, eg, eh

    best, eg, eh = objective(g0, hv0)
    kept, g_cur, trials = [], g0, 0
    while cands and trials < MAX_TRIALS and (eg > EXPLAIN_G / 2 or eh > EXPLAIN_H / 2):
        sreq = [s_.clone().requires_grad_(True) for s_ in sels]
        gf, _ = evaluate(sreq, False, True)
        ds = torch.autograd.grad((gf * (g_e - g_cur)).sum(), sreq, allow_unused=True)
        for cd in cands:
            n, c, y, x = cd['at']
            d = ds[cd['block']]
            if d is None:
                cd['score'] = 0.0
            elif cd['kind'] == 'relu':
                cd['score'] = float(d[n, c, y, x, cd['a']]) * (1.0 - 2.0 * float(sels[cd['block']][n, c, y, x, cd['a']]))
            else:
                cd['score'] = float(d[n, c, y, x, cd['b']] - d[n, c, y, x, cd['a']])
        by_score = [cd for cd in sorted(cands, key=lambda cd: -cd['score']) if cd['score'] > 0.0][:6]
        by_margin = [cd for cd in sorted(cands, key=lambda cd: abs(cd['margin'])) if all(cd is not o for o in by_score)]
        improved = False
        for cd in by_score + by_margin:
            if trials >= MAX_TRIALS:
                break
            trials += 1
            cands = [o for o in cands if o is not cd]
            _apply(sels, cd)
            g1, hv1 = evaluate(sels, hv_e is not None, False)
            j1, eg1, eh1 = objective(g1, hv1)
            if j1 < 0.9 * best:          # a true flip removes its whole contribution; a chance 2 % gain is not an explanation
                best, eg, eh, g_cur = j1, eg1, eh1, g1
                kept.append(dict(block=cd['block'], at=cd['at'], kind=cd['kind'], margin=cd['margin']))
                improved = True
                break
            _apply(sels, cd)          # undo
        if not improved:
            break
    return eg, eh, kept


def explain_step(net, theta_k, shapes, g_e, v, hv_e, g0, hv0):
    """`explain` for one support (or query) pass of the classifier at theta_k."""
    rec = []
    net.forward(_unflatten(theta_k, shapes), None, rec)

    def evaluate(sels, want_hv, graph):
        g, hv, _ = net.grad_hvp(theta_k, shapes, v if want_hv else None, sels, None, keep_graph=graph)
        return g, hv

    return explain(evaluate, rec, g_e, hv_e, g0, hv0)


def anil_task(job):
    """ANIL (reference vision/anil_vision.py:86-99,116-122): job = dict(t, theta_feat, theta_head (flat fp32, reference order),
    grad (engine, [P]), data, labels, shots, ways, K, lr, hidden).  -> dict(t, e64, e32 raw errors of the engine's per-task
    meta-gradient against the two legs, ex = near-tie adjusted, loss64, flips)."""
    import torch
    import torch.nn.functional as F
    from oracle import vision_ref as R
    torch.set_num_threads(int(job.get('threads', 4)))
    ways, shots, K, lr = int(job['ways']), int(job['shots']), int(job['K']), float(job['lr'])
    base = R.convbase_spec(hidden=int(job['hidden']), channels=3, max_pool=True)
    fshapes = R.param_shapes(dict(base=base, in_shape=(3, 84, 84)), '0.', False)
    fc = int(job['hidden']) * 25
    hshapes = OrderedDict([('weight', (ways, fc)), ('bias', (ways,))])
    tf32, th32 = torch.from_numpy(np.asarray(job['theta_feat'])), torch.from_numpy(np.asarray(job['theta_head']))
    g_e = torch.from_numpy(np.asarray(job['grad'])).double()
    data, labels = torch.from_numpy(job['data']), torch.from_numpy(job['labels'])
    out = dict(t=int(job['t']))
    for dt, tag in ((torch.float64, '64'), (torch.float32, '32')):
        tfd = OrderedDict((k, v.to(dt)) for k, v in _unflatten(tf32, fshapes).items())
        thd = OrderedDict((k, v.to(dt)) for k, v in _unflatten(th32, hshapes).items())
        l, a, gf, gh = R.anil_meta_batch(tfd, thd, base, fc, [data.to(dt)], [labels], K, shots, ways, lr, False)
        g = torch.cat([R.flatten_params(gf), R.flatten_params(gh)]).double()
        out['e' + tag] = rel_err(g_e.numpy(), g.numpy())
        out['loss' + tag], out['acc' + tag] = float(l[0]), float(a[0])
        if tag == '64':
            g64 = g
    out['ex'], out['flips'] = out['e64'], []
    if out['e64'] > EXPLAIN_G and bool(job.get('explain', True)):
        net = DecisionNet(data.double(), None, dict(base=base))
        tf64, th64 = tf32.double(), th32.double()
        si, qi = R.prepare_batch_indices(data.shape[0], shots, ways)
        si, qi = torch.from_numpy(si), torch.from_numpy(qi)

        def evaluate(sels, want_hv, graph, record=None):
            pf = OrderedDict((n, t.clone().requires_grad_(True)) for n, t in _unflatten(tf64, fshapes).items())
            ph = OrderedDict((n, t.clone().requires_grad_(True)) for n, t in _unflatten(th64, hshapes).items())
            feats = net.trunk(net.x, pf, base, '0.', sels, record).reshape(-1, fc)
            h = OrderedDict((n, t.clone()) for n, t in ph.items())
            for _ in range(K):
                ls = F.cross_entropy(F.linear(feats[si], h['weight'], h['bias']), labels[si])
                gs = torch.autograd.grad(ls, list(h.values()), create_graph=True)
                h = OrderedDict((n, t - lr * gg) for (n, t), gg in zip(h.items(), gs))
            lq = F.cross_entropy(F.linear(feats[qi], h['weight'], h['bias']), labels[qi])
            g = torch.autograd.grad(lq, list(pf.values()) + list(ph.values()), create_graph=graph)
            gfl = torch.cat([t.reshape(-1) for t in g])
            return (gfl if graph else gfl.detach()), None

        rec = []
        g0, _ = evaluate(None, False, False, rec)
        assert rel_err(g0.numpy(), g64.numpy()) < 1e-9, 'the decision-explicit ANIL restatement left the oracle'
        out['ex'], _, out['flips'] = explain(evaluate, rec, g_e, None, g0, None)
    return out


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


def adapt_task(job):
    """Post-adaptation query loss / accuracy of one task by the reference loop (oracle/vision_ref.py::maml_meta_batch without the
    outer backward; first-order adaptation graph -- the forward values do not depend on the order) in fp64 and fp32."""
    import torch
    from oracle import vision_ref as R
    torch.set_num_threads(int(job.get('threads', 4)))
    ways, shots = int(job['ways']), int(job['shots'])
    spec = R.mini_imagenet_spec(ways)
    shapes = R.param_shapes(spec)
    theta = torch.from_numpy(np.asarray(job['theta']))
    out = dict(t=int(job['t']))
    for dt, tag in ((torch.float64, '64'), (torch.float32, '32')):
        th = OrderedDict((k, v.to(dt)) for k, v in _unflatten(theta, shapes).items())
        l, a, _, _ = R.maml_meta_batch(th, spec, [torch.from_numpy(job['data']).to(dt)], [torch.from_numpy(job['labels'])], int(job['K']),
                                       shots, ways, float(job['lr']), True, backward=False)
        out['loss' + tag], out['acc' + tag] = float(l[0]), float(a[0])
    return out


def adapt_all(theta, data, labels, shots, ways, K, lr, tasks, workers=None, threads=4, timeout=1100):
    jobs = [('adapt', dict(t=int(t), theta=np.asarray(theta), data=data[i], labels=labels[i], shots=shots, ways=ways, K=K, lr=lr,
                           threads=threads)) for i, t in enumerate(tasks)]
    return _run_workers(jobs, workers, threads, timeout)


def _run_workers(jobs, workers, threads, timeout):
    """jobs: list of (kind, dict of numpy arrays / scalars).  Runs them in `python teacher_forced.py jobs.npz out.json` children (the
    parent keeps the GPU; importing torch opens the device, so the children count against the GPU box's 6-process guard: <= 4)."""
    import json
    import pickle
    import subprocess
    import tempfile
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    if workers is None:
        workers = max(1, min(len(jobs), cores // threads, 4))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(workers):
            mine = jobs[w::workers]
            if not mine:
                continue
            job, out = os.path.join(tmp, f'job{w}.pkl'), os.path.join(tmp, f'out{w}.json')
            with open(job, 'wb') as f:
                pickle.dump(mine, f)
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), job, out], env=env), out))
        res = []
        for p, out in procs:
            try:
                rc = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q, _ in procs:
                    q.kill()
                raise RuntimeError('oracle worker timed out')
            if rc != 0:
                raise RuntimeError(f'oracle worker failed (exit {rc})')
            with open(out) as f:
                res += json.load(f)
    return sorted(res, key=lambda r: r['t'])


def teacher_forced_all(trace, data, labels, shots, ways, tasks, workers=None, threads=4, timeout=900, explain=True):
    """`teacher_forced_task` for every task in `tasks` in CPU worker processes.  trace: dict of [*, T, P] tensors from
    MetaEngine.set_trace."""
    tr = {k: trace[k].detach().cpu().numpy() for k in ('theta', 'g', 'lam_in', 'hv')}
    jobs = [('maml', dict(t=int(t), theta=tr['theta'][:, t], g=tr['g'][:, t], lam_in=tr['lam_in'][:, t], hv=tr['hv'][:, t], data=data[t],
                          labels=labels[t], shots=shots, ways=ways, threads=threads, explain=explain)) for t in tasks]
    return _run_workers(jobs, workers, threads, timeout)


def anil_all(theta_feat, theta_head, grads, data, labels, shots, ways, K, lr, hidden, tasks, workers=None, threads=4, timeout=900):
    """`anil_task` for every task in `tasks` (grads: {task: engine per-task meta-gradient [P]})."""
    jobs = [('anil', dict(t=int(t), theta_feat=np.asarray(theta_feat), theta_head=np.asarray(theta_head), grad=np.asarray(grads[t]),
                          data=data[t], labels=labels[t], shots=shots, ways=ways, K=K, lr=lr, hidden=hidden, threads=threads))
            for t in tasks]
    return _run_workers(jobs, workers, threads, timeout)


def _worker_main(job_path, out_path):
    import json
    import pickle
    with open(job_path, 'rb') as f:
        jobs = pickle.load(f)
    res = [dict(maml=teacher_forced_task, anil=anil_task, adapt=adapt_task)[kind](job) for kind, job in jobs]
    with open(out_path, 'w') as f:
        json.dump(res, f)


if __name__ == '__main__':
    _worker_main(sys.argv[1], sys.argv[2])
