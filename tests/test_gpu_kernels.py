"""Per-kernel parity on the GPU: every exported kernel entry point of libmi_maml.so (called through the C ABI) against its
fp64 restatement in oracle/kernels_ref.py on the same seeded inputs.  fp32 kernels vs fp64 oracle: tolerances stated per test."""
import ctypes as C

import numpy as np
import pytest
import torch

from exploring_meta_amd import _lib
from exploring_meta_amd.utils import synthetic
from oracle import kernels_ref as KR
from oracle import vision_ref as R
from gpu_utils import dev, ptr, stream, rel_err, max_err, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['split_f16', 'split_bf16_16x16', 'split_bf16_32x32', 'fp32_pipe'])
def lib(request):
    """Every case runs with both operand forms of the 32-channel stride-1 convolutions (mi_conv_set_split_bf16): same bars."""
    lb = _lib.load()
    from conftest import apply_conv_form
    restore = apply_conv_form(lb, request.param)
    yield lb
    restore()


def _rand(seed, shape, lo=-1.0, hi=1.0):
    return synthetic.hash_uniform(seed, shape) * (hi - lo) + lo


def test_prepare_batch_bit_exact(lib):
    T, ways, shots, c, h, w = 3, 5, 2, 3, 12, 10
    n2 = 2 * ways * shots
    data = _rand(1, (T, n2, c, h, w), 0, 255).astype(np.float32)
    labels = np.stack([synthetic.task_labels(ways, shots) + 0 for _ in range(T)])
    d, l = dev(data), dev(labels, torch.int64)
    xs = torch.empty(T, n2 // 2, h, w, c, device='cuda')
    xq = torch.empty_like(xs)
    ys = torch.empty(T, n2 // 2, dtype=torch.int32, device='cuda')
    yq = torch.empty_like(ys)
    _lib.check(lib.mi_prepare_batch(stream(), ptr(d), ptr(l), T, n2, c, h, w, ptr(xs), ptr(xq), ptr(ys), ptr(yq)))
    si, qi = R.prepare_batch_indices(n2, shots, ways)
    assert np.array_equal(xs.cpu().numpy(), data[:, si].transpose(0, 1, 3, 4, 2))
    assert np.array_equal(xq.cpu().numpy(), data[:, qi].transpose(0, 1, 3, 4, 2))
    assert np.array_equal(ys.cpu().numpy(), labels[:, si]) and np.array_equal(yq.cpu().numpy(), labels[:, qi])


CONV_CASES = [
    # name, T, n, h, w, ci, co, stride
    ('min_l1', 2, 3, 84, 84, 3, 32, 1),
    ('min_l2', 2, 3, 42, 42, 32, 32, 1),
    ('min_l3_odd', 3, 5, 21, 21, 32, 32, 1),
    ('min_l4', 2, 5, 10, 10, 32, 32, 1),
    ('anil_l1', 1, 2, 20, 20, 3, 64, 1),
    ('anil_l2', 2, 2, 21, 21, 64, 64, 1),
    ('omni_l1', 2, 5, 28, 28, 1, 64, 2),
    ('omni_l2', 2, 5, 14, 14, 64, 64, 2),
    ('omni_l3', 2, 5, 7, 7, 64, 64, 2),
    ('omni_l4', 2, 5, 4, 4, 64, 64, 2),
    # maps narrower than a tile and rectangular ones: several image rows (and images) inside one 30- / 32-pixel tile, every lane next to a
    # column boundary (the shifted operands of the split form), partial last tiles
    ('tiny_7x7', 3, 4, 7, 7, 32, 32, 1),
    ('tiny_3x3', 2, 5, 3, 3, 32, 32, 1),
    ('row_1x5', 2, 3, 1, 5, 32, 32, 1),
    ('rect_6x33', 2, 2, 6, 33, 32, 32, 1),
    ('tiny_7x7_64', 2, 3, 7, 7, 64, 64, 1),
]


def _conv_inputs(T, n, h, w, ci, co, seed):
    x = _rand(seed, (T, n, h, w, ci), 0.0, 2.0)
    w9 = _rand(seed + 1, (T, 9, ci, co), -0.3, 0.3)
    return x, w9


@pytest.mark.parametrize('name,T,n,h,w,ci,co,stride', CONV_CASES)
def test_conv_bn_stats(lib, name, T, n, h, w, ci, co, stride):
    x, w9 = _conv_inputs(T, n, h, w, ci, co, 10)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    pstride = 9 * ci * co + 17
    wbuf = np.zeros((T, pstride), np.float32)
    wbuf[:, :9 * ci * co] = w9.reshape(T, -1)
    xd, wd = dev(x), dev(wbuf)
    z = torch.full((T, n, ho, wo, co), float('nan'), device='cuda')
    mu = torch.empty(T, co, device='cuda')
    rstd = torch.empty(T, co, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, co)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    _lib.check(lib.mi_conv3x3_bn_stats(stream(), ptr(xd), ptr(wd), pstride, T, n, h, w, ci, co, stride, ptr(z), ptr(mu),
                                       ptr(rstd), ptr(scratch), sb))
    torch.cuda.synchronize()
    x32, w32 = torch.from_numpy(x.astype(np.float32)).double(), torch.from_numpy(w9.astype(np.float32)).double()
    ez, emu, er = [], [], []
    for t in range(T):
        zr = KR.conv3x3(x32[t], w32[t], stride)
        m, r = KR.bn_stats(zr)
        ez.append(rel_err(z[t].cpu().numpy(), zr.numpy()))
        emu.append(max_err(mu[t].cpu().numpy(), m.numpy()) / max(1e-30, float(zr.std())))
        er.append(rel_err(rstd[t].cpu().numpy(), r.numpy()))
    report(f'conv_bn_stats[{name}]', z_rel=max(ez), mu_err_over_std=max(emu), rstd_rel=max(er))
    assert max(ez) < 2e-6 and max(emu) < 2e-6 and max(er) < 2e-6


BN_CASES = [('pool_even', 2, 3, 42, 42, 32, 1), ('pool_odd', 2, 4, 21, 21, 32, 1), ('pool_c64', 1, 2, 10, 10, 64, 1),
            ('nopool', 2, 5, 14, 14, 64, 0), ('nopool_small', 3, 5, 2, 2, 64, 0)]


@pytest.mark.parametrize('name,T,n,ho,wo,c,pool', BN_CASES)
def test_bn_relu_pool_fwd_bwd(lib, name, T, n, ho, wo, c, pool):
    z = _rand(20, (T, n, ho, wo, c), -2.0, 3.0).astype(np.float32)
    hp, wp = (ho // 2, wo // 2) if pool else (ho, wo)
    gamma = _rand(21, (T, c), 0.1, 1.0).astype(np.float32)
    beta = _rand(22, (T, c), -0.3, 0.3).astype(np.float32)
    dp = _rand(23, (T, n, hp, wp, c)).astype(np.float32)
    pstride = 2 * c + 8
    pb = np.zeros((T, pstride), np.float32)
    pb[:, :c] = gamma
    pb[:, c:2 * c] = beta
    zt = torch.from_numpy(z).double()
    stats = [KR.bn_stats(zt[t]) for t in range(T)]
    mu = np.stack([s[0].numpy() for s in stats]).astype(np.float32)
    rstd = np.stack([s[1].numpy() for s in stats]).astype(np.float32)
    zd, mud, rd, pbd, dpd = dev(z), dev(mu), dev(rstd), dev(pb), dev(dp)
    p = torch.full((T, n, hp, wp, c), float('nan'), device='cuda')
    _lib.check(lib.mi_bn_relu_pool(stream(), ptr(zd), ptr(mud), ptr(rd), ptr(pbd), C.c_void_p(pbd.data_ptr() + 4 * c), pstride,
                                   T, n, ho, wo, c, pool, ptr(p)))
    gb = torch.zeros(T, pstride, device='cuda')
    dz = torch.full((T, n, ho, wo, c), float('nan'), device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, ho, wo, c)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    _lib.check(lib.mi_bn_relu_pool_bwd(stream(), ptr(zd), ptr(mud), ptr(rd), ptr(pbd), C.c_void_p(pbd.data_ptr() + 4 * c),
                                       pstride, ptr(dpd), T, n, ho, wo, c, pool, ptr(gb), C.c_void_p(gb.data_ptr() + 4 * c),
                                       pstride, ptr(dz), ptr(scratch), sb))
    torch.cuda.synchronize()
    ep, eg, eb, edz = [], [], [], []
    for t in range(T):
        m, r = torch.from_numpy(mu[t]).double(), torch.from_numpy(rstd[t]).double()
        g, b = torch.from_numpy(gamma[t]).double(), torch.from_numpy(beta[t]).double()
        pr = KR.bn_relu_pool_fwd(zt[t], m, r, g, b, bool(pool))
        dgr, dbr, dzr = KR.bn_bwd(zt[t], m, r, g, b, torch.from_numpy(dp[t]).double(), bool(pool))
        ep.append(max_err(p[t].cpu().numpy(), pr.numpy()))
        eg.append(rel_err(gb[t, :c].cpu().numpy(), dgr.numpy()))
        eb.append(rel_err(gb[t, c:2 * c].cpu().numpy(), dbr.numpy()))
        edz.append(rel_err(dz[t].cpu().numpy(), dzr.numpy()))
    report(f'bn_relu_pool[{name}]', p_max=max(ep), dgamma_rel=max(eg), dbeta_rel=max(eb), dz_rel=max(edz))
    assert max(ep) < 5e-6 and max(eg) < 5e-6 and max(eb) < 5e-6 and max(edz) < 5e-6


@pytest.mark.parametrize('name,T,n,h,w,ci,co,stride', CONV_CASES)
def test_conv_bwd(lib, name, T, n, h, w, ci, co, stride):
    x, w9 = _conv_inputs(T, n, h, w, ci, co, 30)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    dzv = _rand(31, (T, n, ho, wo, co)).astype(np.float32)
    pstride = 9 * ci * co + 3
    wbuf = np.zeros((T, pstride), np.float32)
    wbuf[:, :9 * ci * co] = w9.reshape(T, -1)
    xd, wd, dzd = dev(x), dev(wbuf), dev(dzv)
    need_dx = ci >= 32
    dx = torch.full((T, n, h, w, ci), float('nan'), device='cuda') if need_dx else None
    dw = torch.full((T, pstride), float('nan'), device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, co)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    _lib.check(lib.mi_conv3x3_bwd(stream(), ptr(xd), ptr(dzd), ptr(wd), pstride, T, n, h, w, ci, co, stride, ptr(dx), ptr(dw),
                                  pstride, ptr(scratch), sb))
    torch.cuda.synchronize()
    x32, w32 = torch.from_numpy(x.astype(np.float32)).double(), torch.from_numpy(w9.astype(np.float32)).double()
    edx, edw = [0.0], []
    for t in range(T):
        dzt = torch.from_numpy(dzv[t]).double()
        dwr = KR.conv3x3_wgrad(x32[t], dzt, stride)
        edw.append(rel_err(dw[t, :9 * ci * co].cpu().numpy(), dwr.numpy()))
        if need_dx:
            dxr = KR.conv3x3_dgrad(dzt, w32[t], (h, w), stride)
            edx.append(rel_err(dx[t].cpu().numpy(), dxr.numpy()))
    report(f'conv_bwd[{name}]', dx_rel=max(edx), dw_rel=max(edw))
    assert max(edx) < 2e-6 and max(edw) < 5e-6


@pytest.mark.parametrize('T,n,feat,ways', [(3, 25, 800, 5), (2, 5, 64, 5), (2, 20, 64, 20)])
def test_head_fwd_bwd(lib, T, n, feat, ways):
    f = _rand(40, (T, n, feat), 0.0, 1.5).astype(np.float32)
    wl = _rand(41, (T, ways, feat), -0.1, 0.1).astype(np.float32)
    bl = _rand(42, (T, ways), -0.1, 0.1).astype(np.float32)
    y = (np.arange(T * n).reshape(T, n) % ways).astype(np.int32)
    pstride = ways * feat + ways + 7
    pb = np.zeros((T, pstride), np.float32)
    pb[:, :ways * feat] = wl.reshape(T, -1)
    pb[:, ways * feat:ways * feat + ways] = bl
    fd, pbd, yd = dev(f), dev(pb), dev(y, torch.int32)
    loss = torch.empty(T, device='cuda')
    acc = torch.empty(T, device='cuda')
    logits = torch.empty(T, n, ways, device='cuda')
    prob = torch.empty_like(logits)
    dl = torch.empty_like(logits)
    gb = torch.zeros(T, pstride, device='cuda')
    df = torch.empty(T, n, feat, device='cuda')
    off_b = 4 * ways * feat
    _lib.check(lib.mi_head_fwd_bwd(stream(), ptr(fd), ptr(pbd), C.c_void_p(pbd.data_ptr() + off_b), pstride, ptr(yd), T, n, feat,
                                   ways, ptr(loss), ptr(acc), ptr(logits), ptr(prob), ptr(dl), ptr(gb),
                                   C.c_void_p(gb.data_ptr() + off_b), pstride, ptr(df)))
    torch.cuda.synchronize()
    errs = dict(loss=0.0, logits=0.0, dwl=0.0, dbl=0.0, df=0.0)
    for t in range(T):
        lr_, ar, lg, pr, dlr, dwr, dbr, dfr = KR.head_fwd_bwd(torch.from_numpy(f[t]).double(), torch.from_numpy(wl[t]).double(),
                                                              torch.from_numpy(bl[t]).double(), torch.from_numpy(y[t]).long())
        errs['loss'] = max(errs['loss'], abs(loss[t].item() - lr_.item()))
        errs['logits'] = max(errs['logits'], max_err(logits[t].cpu().numpy(), lg.numpy()))
        errs['dwl'] = max(errs['dwl'], rel_err(gb[t, :ways * feat].cpu().numpy(), dwr.numpy()))
        errs['dbl'] = max(errs['dbl'], rel_err(gb[t, ways * feat:ways * feat + ways].cpu().numpy(), dbr.numpy()))
        errs['df'] = max(errs['df'], rel_err(df[t].cpu().numpy(), dfr.numpy()))
        assert acc[t].item() == ar.item()
    report(f'head[{T},{n},{feat},{ways}]', **errs)
    assert errs['loss'] < 2e-6 and errs['logits'] < 5e-6 and errs['dwl'] < 5e-6 and errs['dbl'] < 5e-6 and errs['df'] < 5e-6


# ---------------------------------------------------------------------------------------------- block-1 statistics from the input Gram matrix
def _patches(x, ci):
    """x [n,H,W,ci] -> P [n*H*W, 9*ci] zero-padded 3x3 patches, entry a = tap*ci + c (the weight-row order)."""
    n, H, W, _ = x.shape
    xp = np.zeros((n, H + 2, W + 2, ci), np.float64)
    xp[:, 1:-1, 1:-1] = x
    cols = [xp[:, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)]
    return np.concatenate(cols, axis=-1).reshape(n * H * W, 9 * ci)


@pytest.mark.parametrize('name,T,n,h,w,ci,co,lo,hi', [('min', 2, 3, 84, 84, 3, 32, 0.0, 255.0), ('omni_like', 3, 5, 28, 28, 1, 64, 0.0, 1.0),
                                                       ('odd_w', 1, 2, 10, 14, 3, 32, -1.0, 1.0), ('tiny', 2, 1, 4, 6, 1, 32, 0.0, 2.0)])
def test_input_gram_and_stats(lib, name, T, n, h, w, ci, co, lo, hi):
    x = _rand(21, (T, n, h, w, ci), lo, hi).astype(np.float32)
    # weights with a large DC component: |mean z| >> std z, the case where E[z^2] - mean^2 cancels (block1.hip comment)
    w9 = (_rand(22, (T, 9, ci, co), -0.05, 0.05) + 0.3).astype(np.float32)
    w9d = _rand(23, (T, 9, ci, co), -1.0, 1.0).astype(np.float32)
    ng = 32 if ci == 3 else 16
    kp = 9 * ci
    xd = dev(x)
    sb = lib.mi_input_gram_scratch_bytes(T, n, h, ci)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    g = torch.full((T, ng, ng), float('nan'), dtype=torch.float64, device='cuda')
    _lib.check(lib.mi_input_gram(stream(), ptr(xd), T, n, h, w, ci, ptr(scratch), sb, ptr(g)))
    pstride = kp * co + 5
    wbuf, vbuf = np.zeros((T, pstride), np.float32), np.zeros((T, pstride), np.float32)
    wbuf[:, :kp * co], vbuf[:, :kp * co] = w9.reshape(T, -1), w9d.reshape(T, -1)
    wd_, vd_ = dev(wbuf), dev(vbuf)
    mu, rstd, m1, m2 = (torch.empty(T, co, device='cuda') for _ in range(4))
    M = n * h * w
    _lib.check(lib.mi_gram_bn_stats(stream(), ptr(g), T, ci, co, ptr(wd_), pstride, None, 0, M, ptr(mu), ptr(rstd), None, None))
    _lib.check(lib.mi_gram_bn_stats(stream(), ptr(g), T, ci, co, ptr(wd_), pstride, ptr(vd_), pstride, M, ptr(m1), ptr(m2), ptr(mu), ptr(rstd)))
    torch.cuda.synchronize()
    gh = g.cpu().numpy()
    worst = {}
    for t in range(T):
        P = _patches(x[t].astype(np.float64), ci)
        Pe = np.concatenate([P, np.ones((M, 1))], axis=1)
        G = Pe.T @ Pe
        assert np.allclose(gh[t, :kp + 1, :kp + 1], G, rtol=1e-12, atol=1e-9 * np.abs(G).max())
        assert np.all(gh[t, kp + 1:, :] == 0.0) and np.all(gh[t, :, kp + 1:] == 0.0)
        z = P @ w9[t].reshape(kp, co).astype(np.float64)
        zd = P @ w9d[t].reshape(kp, co).astype(np.float64)
        mean, var = z.mean(0), z.var(0)
        r = 1.0 / np.sqrt(var + 1e-5)
        mu_t, r_t = mu[t].cpu().numpy().astype(np.float64), rstd[t].cpu().numpy().astype(np.float64)
        zh = (z - mu_t) * r_t
        for k, (got, want, scale) in {'mu': (mu_t, mean, np.sqrt(var)), 'rstd': (r_t, r, r), 'm1': (m1[t].cpu().numpy(), zd.mean(0), zd.std(0)),
                                      'm2': (m2[t].cpu().numpy(), (zh * zd).mean(0), zd.std(0))}.items():
            worst[k] = max(worst.get(k, 0.0), float(np.max(np.abs(got - want) / scale)))
    report(f'gram_stats[{name}]', **worst)
    assert worst['mu'] < 1e-6 and worst['rstd'] < 1e-6 and worst['m1'] < 1e-6 and worst['m2'] < 1e-6


@pytest.mark.parametrize('sx,sw,spread', [(1.0, 1.0, True), (1e15, 1e15, False), (1e-18, 1e-18, False), (1e18, 1e-18, False)])
def test_conv_operand_forms_keep_fp32_accuracy_across_magnitudes(lib, sx, sw, spread):
    """The hidden convolution on inputs and weights from 1e-18 to 1e18 (and, `spread`, element magnitudes spread over twelve decades
    inside one tensor): both operand forms stay at fp32 rounding of the fp64 result -- the three-way bf16 split is exact at every exponent
    (bf16 has fp32's exponent range), the dropped cross terms stay 2^-24 of a product."""
    T, n, h, w, c = 2, 3, 21, 21, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, n, h, w, c, generator=g) * sx
    if spread:
        x = x * torch.pow(10.0, torch.randint(-6, 7, x.shape, generator=g).float())
    ps = 9 * c * c + 64
    wb = torch.randn(T, ps, generator=g) * sw
    xd, wd = x.cuda(), wb.cuda()
    z = torch.full((T, n, h, w, c), float('nan'), device='cuda')
    mu, rstd = torch.empty(T, c, device='cuda'), torch.empty(T, c, device='cuda')
    sb = lib.mi_kernel_scratch_bytes(T, n, h, w, c)
    scratch = torch.empty(sb, dtype=torch.uint8, device='cuda')
    _lib.check(lib.mi_conv3x3_bn_stats(stream(), ptr(xd), ptr(wd), ps, T, n, h, w, c, c, 1, ptr(z), ptr(mu), ptr(rstd), ptr(scratch), sb))
    torch.cuda.synchronize()
    for t in range(T):
        ref = KR.conv3x3(x[t].double(), wb[t, :9 * c * c].reshape(9, c, c).double())
        assert torch.isfinite(z[t]).all()
        assert rel_err(z[t].cpu().numpy(), ref.numpy()) < 1e-6
