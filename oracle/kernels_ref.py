"""Oracle (TEST INFRASTRUCTURE): explicit per-kernel restatement of the hot path in the ENGINE's decomposition.

``vision_ref.py`` follows the reference literally (autograd, reverse-over-reverse).  This file restates the same
mathematics the way the HIP engine computes it -- NHWC activations, tap-major conv weights, explicit backward formulas,
and the second-order meta-gradient by the adjoint recursion  lam_k = lam_{k+1} - alpha * H_s(theta_k) lam_{k+1}  with each
Hessian-vector product done forward-over-reverse (tangent pass) -- so every HIP kernel has a CPU counterpart with the
same inputs/outputs.  ``tests/test_oracle_kernels.py`` proves it equal to ``vision_ref`` (hence to the reference goldens).

Per ConvBlock (reference vision_models.py:188-193), z = conv(x, W) [+ b, which batch-stat BN cancels exactly]:
    mu, var (biased) over (N,H,W);  r = 1/sqrt(var+eps);  zh = (z-mu) r;  u = gamma zh + beta;  a = relu(u);  p = maxpool2(a)
backward (M = N*H*W):   du = route(dp) * [u>0];  dgamma = sum du zh;  dbeta = sum du;
                        dz = gamma r (du - dbeta/M - zh dgamma/M);  dW = wgrad(x, dz);  dx = dgrad(dz, W)
tangent (R-operator, direction = parameter tangents, input tangent xd):
    zd = conv(xd, W) + conv(x, Wd);  m1 = mean zd;  m2 = mean(zh zd);  zhd = r (zd - m1 - zh m2);  rd = -r^2 m2
    ud = gammad zh + gamma zhd + betad;  pd = ud at the pooling argmax where u>0
    R{dbeta} = sum dud;  R{dgamma} = sum(dud zh + du zhd)
    R{dz} = (gammad r + gamma rd) E + gamma r (dud - R{dbeta}/M - zhd dgamma/M - zh R{dgamma}/M),  E = du - dbeta/M - zh dgamma/M
    R{dW} = wgrad(xd, dz) + wgrad(x, R{dz});   R{dx} = dgrad(R{dz}, W) + dgrad(dz, Wd)
"""

from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1e-5


# ------------------------------------------------------------------------------------------ layouts
def net_desc(spec):
    """Layer table of a reference model spec (oracle.vision_ref.*_spec): dicts with ci, co, h, w (input), ho, wo (conv
    output), stride, pool, hp, wp (block output)."""
    b = spec['base'] if 'base' in spec else spec
    c, h, w = spec['in_shape']
    layers = []
    ci = b['channels']
    s = int(2 * b['max_pool_factor'])
    for _ in range(b['layers']):
        stride = 1 if b['max_pool'] else s
        ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        hp, wp = (ho // s, wo // s) if b['max_pool'] else (ho, wo)
        layers.append(dict(ci=ci, co=b['hidden'], h=h, w=w, ho=ho, wo=wo, stride=stride, pool=bool(b['max_pool']), hp=hp, wp=wp))
        ci, h, w = b['hidden'], hp, wp
    head = None
    if 'ways' in spec:
        head = dict(ways=spec['ways'], mean_pool=(spec['kind'] == 'omni'), c=ci, hw=h * w,
                    f=(ci if spec['kind'] == 'omni' else ci * h * w))
    return dict(layers=layers, head=head, in_shape=(c, spec['in_shape'][1], spec['in_shape'][2]))


def to_engine_params(p, desc, prefix='base.'):
    """Reference-named params -> engine layout list: per layer (gamma, beta, W[9,ci,co], b), then (Wl[ways,F_nhwc], bl)."""
    out = []
    for i, L in enumerate(desc['layers']):
        w = p[f'{prefix}{i}.conv.weight']                       # [co, ci, 3, 3]
        out += [p[f'{prefix}{i}.normalize.weight'], p[f'{prefix}{i}.normalize.bias'],
                w.permute(2, 3, 1, 0).reshape(9, L['ci'], L['co']).contiguous(), p[f'{prefix}{i}.conv.bias']]
    if desc['head'] is not None and 'linear.weight' in p:
        hd = desc['head']
        wl = p['linear.weight']
        if not hd['mean_pool']:                                  # NCHW flatten c*HW+s  ->  NHWC flatten s*C+c
            wl = wl.reshape(hd['ways'], hd['c'], hd['hw']).permute(0, 2, 1).reshape(hd['ways'], -1).contiguous()
        out += [wl, p['linear.bias']]
    return out


def from_engine_grads(g, desc, names, prefix='base.'):
    """Inverse of ``to_engine_params`` for gradient lists -> OrderedDict with reference names/shapes."""
    out = OrderedDict()
    k = 0
    for i, L in enumerate(desc['layers']):
        out[f'{prefix}{i}.normalize.weight'] = g[k]
        out[f'{prefix}{i}.normalize.bias'] = g[k + 1]
        out[f'{prefix}{i}.conv.weight'] = g[k + 2].reshape(3, 3, L['ci'], L['co']).permute(3, 2, 0, 1).contiguous()
        out[f'{prefix}{i}.conv.bias'] = g[k + 3]
        k += 4
    if desc['head'] is not None and len(g) > k:
        hd = desc['head']
        wl = g[k]
        if not hd['mean_pool']:
            wl = wl.reshape(hd['ways'], hd['hw'], hd['c']).permute(0, 2, 1).reshape(hd['ways'], -1).contiguous()
        out['linear.weight'] = wl
        out['linear.bias'] = g[k + 1]
    return OrderedDict((n, out[n]) for n in names) if names is not None else out


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------------ conv kernels
def _w_oihw(w9, ci, co):
    return w9.reshape(3, 3, ci, co).permute(3, 2, 0, 1)


def conv3x3(x, w9, stride=1):
    """x [N,H,W,Ci], w9 [9,Ci,Co] -> z [N,Ho,Wo,Co]; pad 1, no bias."""
    ci, co = w9.shape[1], w9.shape[2]
    z = F.conv2d(x.permute(0, 3, 1, 2), _w_oihw(w9, ci, co), None, stride=stride, padding=1)
    return z.permute(0, 2, 3, 1).contiguous()


def conv3x3_dgrad(dz, w9, in_hw, stride=1):
    """dx [N,H,W,Ci] = transpose-conv of dz with w9."""
    ci, co = w9.shape[1], w9.shape[2]
    n = dz.shape[0]
    dx = torch.nn.grad.conv2d_input((n, ci, in_hw[0], in_hw[1]), _w_oihw(w9, ci, co), dz.permute(0, 3, 1, 2),
                                    stride=stride, padding=1)
    return dx.permute(0, 2, 3, 1).contiguous()


def conv3x3_wgrad(x, dz, stride=1):
    """dW [9,Ci,Co] = sum over pixels of x(shifted) (x) dz."""
    ci, co = x.shape[3], dz.shape[3]
    dw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (co, ci, 3, 3), dz.permute(0, 3, 1, 2), stride=stride, padding=1)
    return dw.permute(2, 3, 1, 0).reshape(9, ci, co).contiguous()


# ------------------------------------------------------------------------------------------ BN + ReLU + pool
def bn_stats(z):
    m = z.mean(dim=(0, 1, 2))
    var = ((z - m) ** 2).mean(dim=(0, 1, 2))
    return m, 1.0 / torch.sqrt(var + EPS)


def _windows(t, hp, wp):
    """[N,H,W,C] -> [N,hp,wp,4,C] (window order (0,0),(0,1),(1,0),(1,1)); floor pooling drops odd last row/col."""
    n, _, _, c = t.shape
    t = t[:, :2 * hp, :2 * wp].reshape(n, hp, 2, wp, 2, c).permute(0, 1, 3, 2, 4, 5)
    return t.reshape(n, hp, wp, 4, c)


def _unwindows(tw, h, w):
    n, hp, wp, _, c = tw.shape
    full = torch.zeros(n, h, w, c, dtype=tw.dtype)
    full[:, :2 * hp, :2 * wp] = tw.reshape(n, hp, wp, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * hp, 2 * wp, c)
    return full


def _route(z, mu, r, gamma, beta, pool):
    """u, zh and the selection mask sel = [position is its window's (first) argmax] * [u>0]  (all ones*[u>0] without pool)."""
    zh = (z - mu) * r
    u = gamma * zh + beta
    if not pool:
        return u, zh, (u > 0).to(z.dtype)
    n, h, w, c = z.shape
    hp, wp = h // 2, w // 2
    uw = _windows(u, hp, wp)
    arg = uw.argmax(dim=3, keepdim=True)                  # first maximal index
    onehot = torch.zeros_like(uw).scatter_(3, arg, 1.0)
    sel = _unwindows(onehot * (uw > 0).to(z.dtype), h, w)
    return u, zh, sel


def bn_relu_pool_fwd(z, mu, r, gamma, beta, pool):
    u, _, sel = _route(z, mu, r, gamma, beta, pool)
    a = u * sel
    if not pool:
        return a
    hp, wp = z.shape[1] // 2, z.shape[2] // 2
    return _windows(a, hp, wp).sum(dim=3)                 # exactly one selected position per window (or none -> 0)


def _spread(dp, shape, pool):
    """Pooled-grid gradient broadcast back onto every position of its window (zeros on dropped odd edge)."""
    if not pool:
        return dp
    n, h, w, c = shape
    hp, wp = h // 2, w // 2
    return _unwindows(dp.unsqueeze(3).expand(n, hp, wp, 4, c), h, w)


def bn_bwd(z, mu, r, gamma, beta, dp, pool):
    """-> dgamma, dbeta, dz."""
    m = z.shape[0] * z.shape[1] * z.shape[2]
    _, zh, sel = _route(z, mu, r, gamma, beta, pool)
    du = _spread(dp, z.shape, pool) * sel
    dbeta = du.sum(dim=(0, 1, 2))
    dgamma = (du * zh).sum(dim=(0, 1, 2))
    dz = gamma * r * (du - dbeta / m - zh * dgamma / m)
    return dgamma, dbeta, dz


def bn_tangent_fwd(z, zd, mu, r, gamma, beta, gammad, betad, pool):
    """-> pd, (m1, m2)."""
    _, zh, sel = _route(z, mu, r, gamma, beta, pool)
    m1 = zd.mean(dim=(0, 1, 2))
    m2 = (zh * zd).mean(dim=(0, 1, 2))
    zhd = r * (zd - m1 - zh * m2)
    ud = gammad * zh + gamma * zhd + betad
    ad = ud * sel
    if not pool:
        return ad, (m1, m2)
    hp, wp = z.shape[1] // 2, z.shape[2] // 2
    return _windows(ad, hp, wp).sum(dim=3), (m1, m2)


def bn_tangent_bwd(z, zd, mu, r, m1, m2, gamma, beta, gammad, betad, dp, dpd, dgamma, dbeta, pool):
    """-> R{dgamma}, R{dbeta}, R{dz}."""
    m = z.shape[0] * z.shape[1] * z.shape[2]
    _, zh, sel = _route(z, mu, r, gamma, beta, pool)
    zhd = r * (zd - m1 - zh * m2)
    rd = -r * r * m2
    du = _spread(dp, z.shape, pool) * sel
    dud = _spread(dpd, z.shape, pool) * sel
    rdbeta = dud.sum(dim=(0, 1, 2))
    rdgamma = (dud * zh + du * zhd).sum(dim=(0, 1, 2))
    e = du - dbeta / m - zh * dgamma / m
    rdz = (gammad * r + gamma * rd) * e + gamma * r * (dud - rdbeta / m - zhd * dgamma / m - zh * rdgamma / m)
    return rdgamma, rdbeta, rdz


# ------------------------------------------------------------------------------------------ head
def head_features(p_last, hd):
    n = p_last.shape[0]
    if hd['mean_pool']:
        return p_last.reshape(n, hd['hw'], hd['c']).mean(dim=1)
    return p_last.reshape(n, -1)


def head_features_bwd(df, p_shape, hd):
    n = p_shape[0]
    if hd['mean_pool']:
        return (df / hd['hw']).reshape(n, 1, 1, hd['c']).expand(p_shape).contiguous()
    return df.reshape(p_shape)


def head_fwd_bwd(f, wl, bl, y):
    """Linear + CrossEntropy(mean) forward and backward.  -> loss, acc, logits, prob, dl, dwl, dbl, df."""
    n = f.shape[0]
    logits = f @ wl.t() + bl
    lse = torch.logsumexp(logits, dim=1)
    loss = (lse - logits.gather(1, y[:, None])[:, 0]).mean()
    prob = torch.softmax(logits, dim=1)
    acc = (logits.argmax(dim=1) == y).sum().to(torch.float32) / n
    dl = (prob - F.one_hot(y, logits.shape[1]).to(f.dtype)) / n
    return loss, acc, logits, prob, dl, dl.t() @ f, dl.sum(dim=0), dl @ wl


def head_tangent(f, fd, wl, bl, wld, bld, prob, dl):
    """-> R{dwl}, R{dbl}, R{df}."""
    n = f.shape[0]
    ld = fd @ wl.t() + f @ wld.t() + bld
    probd = prob * (ld - (prob * ld).sum(dim=1, keepdim=True))
    rdl = probd / n
    return rdl.t() @ f + dl.t() @ fd, rdl.sum(dim=0), rdl @ wl + dl @ wld


# ------------------------------------------------------------------------------------------ whole-net passes
def net_forward_backward(theta, x, y, desc, need_dx0=False):
    """One support/query pass: loss, acc, gradient list (engine layout) and everything the tangent pass re-uses."""
    saved = dict(x=[x], z=[], mu=[], r=[], dp=[None] * len(desc['layers']), dz=[None] * len(desc['layers']))
    h = x
    for i, L in enumerate(desc['layers']):
        gamma, beta, w9, _ = theta[4 * i:4 * i + 4]
        z = conv3x3(h, w9, L['stride'])
        mu, r = bn_stats(z)
        h = bn_relu_pool_fwd(z, mu, r, gamma, beta, L['pool'])
        saved['z'].append(z), saved['mu'].append(mu), saved['r'].append(r), saved['x'].append(h)
    hd = desc['head']
    wl, bl = theta[-2], theta[-1]
    f = head_features(h, hd)
    loss, acc, logits, prob, dl, dwl, dbl, df = head_fwd_bwd(f, wl, bl, y)
    saved.update(f=f, prob=prob, dl=dl, logits=logits)
    grads = [None] * len(theta)
    grads[-2], grads[-1] = dwl, dbl
    dp = head_features_bwd(df, h.shape, hd)
    for i in reversed(range(len(desc['layers']))):
        L = desc['layers'][i]
        gamma, beta, w9, b = theta[4 * i:4 * i + 4]
        dgamma, dbeta, dz = bn_bwd(saved['z'][i], saved['mu'][i], saved['r'][i], gamma, beta, dp, L['pool'])
        saved['dp'][i], saved['dz'][i] = dp, dz
        grads[4 * i], grads[4 * i + 1] = dgamma, dbeta
        grads[4 * i + 2] = conv3x3_wgrad(saved['x'][i], dz, L['stride'])
        grads[4 * i + 3] = torch.zeros_like(b)            # conv bias is inert under batch-stat BN: sum(dz) == 0
        if i > 0 or need_dx0:
            dp = conv3x3_dgrad(dz, w9, (L['h'], L['w']), L['stride'])
    saved['g'] = grads
    return loss, acc, grads, saved


def net_hvp(theta, saved, v, desc):
    """R_v{grad L}(theta): Hessian-vector product by a tangent forward + tangent backward sweep over the saved pass."""
    nl = len(desc['layers'])
    xd = None                                             # input tangent is zero
    zds, ms, xds = [], [], [None]
    for i, L in enumerate(desc['layers']):
        gamma, beta, w9, _ = theta[4 * i:4 * i + 4]
        gammad, betad, w9d, _ = v[4 * i:4 * i + 4]
        zd = conv3x3(saved['x'][i], w9d, L['stride'])
        if xd is not None:
            zd = zd + conv3x3(xd, w9, L['stride'])
        xd, m = bn_tangent_fwd(saved['z'][i], zd, saved['mu'][i], saved['r'][i], gamma, beta, gammad, betad, L['pool'])
        zds.append(zd), ms.append(m), xds.append(xd)
    hd = desc['head']
    fd = head_features(xd, hd)
    rdwl, rdbl, rdf = head_tangent(saved['f'], fd, theta[-2], theta[-1], v[-2], v[-1], saved['prob'], saved['dl'])
    out = [None] * len(theta)
    out[-2], out[-1] = rdwl, rdbl
    dpd = head_features_bwd(rdf, saved['x'][nl].shape, hd)
    for i in reversed(range(nl)):
        L = desc['layers'][i]
        gamma, beta, w9, b = theta[4 * i:4 * i + 4]
        gammad, betad, w9d, _ = v[4 * i:4 * i + 4]
        rdgamma, rdbeta, rdz = bn_tangent_bwd(saved['z'][i], zds[i], saved['mu'][i], saved['r'][i], ms[i][0], ms[i][1],
                                              gamma, beta, gammad, betad, saved['dp'][i], dpd,
                                              saved['g'][4 * i], saved['g'][4 * i + 1], L['pool'])
        out[4 * i], out[4 * i + 1] = rdgamma, rdbeta
        rdw = conv3x3_wgrad(saved['x'][i], rdz, L['stride'])
        if xds[i] is not None:
            rdw = rdw + conv3x3_wgrad(xds[i], saved['dz'][i], L['stride'])
        out[4 * i + 2] = rdw
        out[4 * i + 3] = torch.zeros_like(b)
        if i > 0:
            dpd = conv3x3_dgrad(rdz, w9, (L['h'], L['w']), L['stride']) + \
                  conv3x3_dgrad(saved['dz'][i], w9d, (L['h'], L['w']), L['stride'])
    return out


def maml_task(theta0, xs, ys, xq, yq, desc, steps, alpha, first_order):
    """One task of the meta-batch (engine algorithm).  theta0: engine-layout list.  -> loss, acc, meta-grad list, logits."""
    thetas, saves = [theta0], []
    for _ in range(steps):
        _, _, g, sv = net_forward_backward(thetas[-1], xs, ys, desc)
        saves.append(sv)
        thetas.append([t - alpha * gi for t, gi in zip(thetas[-1], g)])
    loss, acc, lam, svq = net_forward_backward(thetas[-1], xq, yq, desc)
    if not first_order:
        for k in reversed(range(steps)):
            hv = net_hvp(thetas[k], saves[k], lam, desc)
            lam = [l - alpha * h for l, h in zip(lam, hv)]
    return loss, acc, lam, svq['logits']


# ------------------------------------------------------------------------------------------ fused block-1 bookkeeping
def pool_argmax(z, mu, r, gamma, beta):
    """What the fused block-1 forward kernel keeps per pooling window and channel: the (first-max) argmax position 0..3 of
    u = gamma*zhat + beta, 4 where the maximum did not pass the ReLU, and zhat at that position.  -> arg [N,hp,wp,C] int64,
    zh_at [N,hp,wp,C]."""
    zh = (z - mu) * r
    u = gamma * zh + beta
    hp, wp = z.shape[1] // 2, z.shape[2] // 2
    uw, zw = _windows(u, hp, wp), _windows(zh, hp, wp)
    arg = uw.argmax(dim=3, keepdim=True)
    on = uw.gather(3, arg)[:, :, :, 0] > 0
    zh_at = zw.gather(3, arg)[:, :, :, 0]
    arg = torch.where(on, arg[:, :, :, 0], torch.full_like(arg[:, :, :, 0], 4))
    return arg, zh_at


def at_argmax(t, arg):
    """Value of the full-resolution tensor t [N,H,W,C] at every window's stored argmax (0 where arg == 4)."""
    hp, wp = arg.shape[1], arg.shape[2]
    tw = _windows(t, hp, wp)
    v = tw.gather(3, arg.clamp(max=3).unsqueeze(3))[:, :, :, 0]
    return torch.where(arg < 4, v, torch.zeros_like(v))
