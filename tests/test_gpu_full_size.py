"""Parity at the sizes bench.py times (BASELINE configs at full T), through the C ABI.

Why the checks are per step: with 5 inner steps at lr 0.5 the map theta_0 -> meta-gradient amplifies a 1e-7 perturbation to
1e-2..4e-1 (the reference's own fp32 run deviates from its fp64 run by 2e-1..4e-1 in the meta-gradient on these inputs --
measured with oracle/vision_ref.py, numbers in DESIGN.md section 7), so an end-to-end comparison cannot tell a correct
kernel from one with a 10 % error in the fifth Hessian-vector product.  The engine therefore dumps its per-step state
(mi_debug_set_trace: theta_k, g_k, the vector fed to every Hessian-vector product and its result) and the fp64 oracle is
TEACHER-FORCED: evaluated at the engine's own theta_k.  Each step is then a single forward/backward (or one
Hessian-vector product): the typical (median) step must agree to 1e-5 (1e-4 for the Hessian-vector products) and NO step may
deviate by more than 1e-3 (5e-3).  Measured: steps without a discrete flip agree to 3e-7; a step where one max-pool argmax /
ReLU decision of a near-tied window resolves differently in fp32 and fp64 (the objective is only piecewise smooth) sits at
2e-5..8e-4, the same size as the reference's own fp32-vs-fp64 difference at that theta (reported next to it).  A kernel error
of 10 % in one Hessian-vector product -- what the end-to-end bar cannot see -- is 100x above the max bound; systematic kernel
errors below it are the business of the per-kernel tests (tests/test_gpu_tangent_kernels.py, 1e-6)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from exploring_meta_amd.engine import MetaEngine, ModelSpec
from exploring_meta_amd.utils import synthetic
from oracle import vision_ref as R
from gpu_utils import rel_err, report

pytestmark = pytest.mark.gpu


def _ref_theta(spec, seed=11):
    return OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic.ref_init_weights(R.param_shapes(spec), seed).items())


def _unflatten(flat, shapes):
    out, off = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def _support_loss(spec, p, xs, ys):
    return F.cross_entropy(R.model_forward(xs, p, spec), ys)


def _teacher_forced(spec, shapes, trace, t, K, data, labels, shots, ways, second_order=True):
    """fp64 oracle evaluated at the engine's own per-step state of task t.  -> per-step errors."""
    xs, ys, xq, yq = R.prepare_batch(torch.from_numpy(data[t]).double(), torch.from_numpy(labels[t]), shots, ways)
    eg, eh = [], []
    for k in range(K):
        p = OrderedDict((n, v.clone().requires_grad_(True)) for n, v in _unflatten(trace['theta'][k, t].double().cpu(), shapes).items())
        loss = _support_loss(spec, p, xs, ys)
        g = torch.autograd.grad(loss, list(p.values()), create_graph=second_order)
        eg.append(rel_err(trace['g'][k, t].cpu().numpy(), torch.cat([x.detach().reshape(-1) for x in g]).numpy()))
        if second_order:
            v = _unflatten(trace['lam_in'][k, t].double().cpu(), shapes)
            dot = sum((gi * v[n]).sum() for gi, n in zip(g, p))
            hv = torch.autograd.grad(dot, list(p.values()))
            eh.append(rel_err(trace['hv'][k, t].cpu().numpy(), torch.cat([x.reshape(-1) for x in hv]).numpy()))
    # query pass at theta_K: loss, accuracy and (second order: the first lam_in) its gradient
    pK = OrderedDict((n, v.clone().requires_grad_(True)) for n, v in _unflatten(trace['theta'][K, t].double().cpu(), shapes).items())
    logits = R.model_forward(xq, pK, spec)
    lq = F.cross_entropy(logits, yq)
    gq = torch.cat([x.reshape(-1) for x in torch.autograd.grad(lq, list(pK.values()))])
    return eg, eh, float(lq), float(R.accuracy(logits, yq)), gq.numpy()


def _teacher_forced_ref_fp32(spec, shapes, trace, t, K, data, labels, shots, ways):
    """The reference arithmetic (autograd) in fp32 against itself in fp64 at the engine's theta_k: per-step gradient / HVP deviation."""
    xs, ys, _, _ = R.prepare_batch(torch.from_numpy(data[t]).double(), torch.from_numpy(labels[t]), shots, ways)
    rg, rh = [], []
    for k in range(K):
        res = []
        for dt in (torch.float64, torch.float32):
            p = OrderedDict((n, v.to(dt).clone().requires_grad_(True)) for n, v in _unflatten(trace['theta'][k, t].cpu(), shapes).items())
            g = torch.autograd.grad(_support_loss(spec, p, xs.to(dt), ys), list(p.values()), create_graph=True)
            v = _unflatten(trace['lam_in'][k, t].cpu().to(dt), shapes)
            hv = torch.autograd.grad(sum((gi * v[n]).sum() for gi, n in zip(g, p)), list(p.values()))
            res.append((torch.cat([x.detach().reshape(-1) for x in g]).double().numpy(), torch.cat([x.reshape(-1) for x in hv]).double().numpy()))
        rg.append(rel_err(res[1][0], res[0][0]))
        rh.append(rel_err(res[1][1], res[0][1]))
    return rg, rh


def test_cfg2_T32_teacher_forced_per_step():
    """BASELINE config 2 exactly as benchmarked (32 tasks, 5-way 5-shot, K = 5, lr 0.5, second order, the bench's synthetic
    tasks and initial parameters): every inner-step gradient and every Hessian-vector product of three tasks against the fp64
    oracle at the engine's own theta_k; the theta recursion and the adjoint recursion themselves are checked exactly."""
    ways, shots, K, lr, T = 5, 5, 5, 0.5, 32
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    shapes = R.param_shapes(spec)
    theta = R.flatten_params(_ref_theta(spec, 42)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    eng = MetaEngine(mspec)
    trace = eng.set_trace(T, K)
    loss, acc, grad, _ = eng.meta_batch(theta, torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda(), shots, K, lr)
    torch.cuda.synchronize()
    eng.set_trace(0)
    th, g, lam_in, hv = (trace[k].double() for k in ('theta', 'g', 'lam_in', 'hv'))
    # the two recursions, exactly as the algorithm states them (fp32 axpy: agreement to rounding)
    for k in range(K):
        assert torch.allclose(th[k + 1], th[k] - lr * g[k], rtol=0, atol=2e-6 * float(th[k].abs().max()))
    for k in range(K - 1):                     # lam_k = lam_{k+1} - lr * H_k lam_{k+1} is the vector fed to step k-1
        want = lam_in[k + 1] - lr * hv[k + 1]
        assert rel_err(lam_in[k].cpu().numpy(), want.cpu().numpy()) < 1e-6
    final = (lam_in[0] - lr * hv[0]).sum(dim=0)
    assert rel_err(grad.double().cpu().numpy(), final.cpu().numpy()) < 1e-6, 'meta-gradient != sum over tasks of the last adjoint'
    all_g, all_h, all_q = [], [], []
    for t in (0, 13, 31):
        eg, eh, lq, aq, gq = _teacher_forced(spec, shapes, trace, t, K, data, labels, shots, ways)
        eq = rel_err(lam_in[K - 1, t].cpu().numpy(), gq)
        extra = {}
        if t == 0:        # the reference arithmetic in fp32 at the same theta_k: the size of a discrete flip, for the record
            rg, rh = _teacher_forced_ref_fp32(spec, shapes, trace, t, K, data, labels, shots, ways)
            extra = dict(ref_fp32_vs_fp64_grad_rel_per_step=rg, ref_fp32_vs_fp64_hvp_rel_per_step=rh)
        report(f'cfg2_T32_teacher_forced[task {t}]', grad_rel_per_step=eg, hvp_rel_per_step=eh, query_grad_rel=eq,
               loss=float(loss[t]), loss_oracle=lq, **extra)
        all_g += eg
        all_h += eh
        all_q.append(eq)
        assert abs(float(loss[t]) - lq) <= 1e-5 * max(1.0, abs(lq))
        assert float(acc[t]) == aq
    assert np.median(all_g) < 1e-5 and max(all_g) < 1e-3, all_g
    assert np.median(all_h) < 1e-4 and max(all_h) < 5e-3, all_h
    assert np.median(all_q) < 1e-5 and max(all_q) < 1e-3, all_q


def test_cfg2_T32_batched_vs_one_task_at_a_time():
    """The batched launch against the same engine looped over single tasks (different launch geometry: one tile per wave,
    other partial-sum groupings).  Step 0 -- one forward/backward from identical parameters -- must agree to rounding for
    every task; after that the fp32 difference in partial-sum order (1e-7) is amplified by the chaotic inner loop exactly as
    the reference's own 1-thread-vs-8-thread difference is, so later steps are held to the teacher-forced
    test above and only reported here."""
    ways, shots, K, lr, T = 5, 5, 5, 0.5, 32
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    theta = R.flatten_params(_ref_theta(spec, 42)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(mspec)
    trace = eng.set_trace(T, K)
    loss, acc, grad, _ = eng.meta_batch(theta, d, l, shots, K, lr)
    torch.cuda.synchronize()
    g0, th1 = trace['g'][0].clone(), trace['theta'][1].clone()
    tr1 = eng.set_trace(1, K)
    e0, dl, accs_equal = [], [], 0
    for t in range(T):
        l1, a1, g1, _ = eng.meta_batch(theta, d[t:t + 1], l[t:t + 1], shots, K, lr)
        torch.cuda.synchronize()
        e0.append(rel_err(tr1['g'][0, 0].cpu().numpy(), g0[t].cpu().numpy()))
        assert torch.allclose(tr1['theta'][1, 0], th1[t], rtol=0, atol=1e-5 * float(th1[t].abs().max()))
        dl.append(abs(float(l1[0]) - float(loss[t])) / abs(float(loss[t])))
        accs_equal += int(float(a1[0]) == float(acc[t]))
    eng.set_trace(0)
    report('cfg2_T32_batched_vs_looped', step0_grad_rel_max=max(e0), final_loss_rel_median=float(np.median(dl)),
           final_loss_rel_max=max(dl), acc_equal=accs_equal)
    assert max(e0) < 1e-5
    assert accs_equal >= T - 4 and float(np.median(dl)) < 5e-2


@pytest.mark.parametrize('T', [32, 256])
def test_cfg4_full_T_batched_looped_oracle(T):
    """BASELINE config 4 (5-way 1-shot, one second-order step, 32 tasks per GPU; 256 = the whole meta-batch on one GPU) with
    the reference's initialisers and plateau-free inputs, the well-conditioned setting SURVEY.md 8c calibrated at <= 1e-4:
    per-task meta-gradients (from the trace) batched vs looped vs fp64 oracle."""
    ways, shots, K, lr = 5, 1, 1, 0.5
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    th64 = _ref_theta(spec, 11)
    theta = R.flatten_params(th64).float().cuda().contiguous()
    data, labels = synthetic.make_uniform_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(mspec)
    trace = eng.set_trace(T, K)
    loss, acc, grad, _ = eng.meta_batch(theta, d, l, shots, K, lr)
    torch.cuda.synchronize()
    per_task = (trace['lam_in'][0].double() - lr * trace['hv'][0].double()).cpu()
    assert rel_err(grad.double().cpu().numpy(), per_task.sum(dim=0).numpy()) < 1e-6
    tr1 = eng.set_trace(1, K)
    eg, el = [], []
    check = sorted(set(range(0, T, max(1, T // 16))) | {T - 1})
    for t in check:
        l1, a1, g1, _ = eng.meta_batch(theta, d[t:t + 1], l[t:t + 1], shots, K, lr)
        torch.cuda.synchronize()
        eg.append(rel_err(per_task[t].numpy(), g1.double().cpu().numpy()))
        el.append(abs(float(l1[0]) - float(loss[t])) / abs(float(loss[t])))
        assert float(a1[0]) == float(acc[t])
    eng.set_trace(0)
    eo, lo = [], []
    for t in (0, T // 2, T - 1):
        l64, a64, g64, _ = R.maml_meta_batch(th64, spec, [torch.from_numpy(data[t]).double()], [torch.from_numpy(labels[t])], K, shots,
                                             ways, lr, False)
        eo.append(rel_err(per_task[t].numpy(), R.flatten_params(g64).numpy()))
        lo.append(abs(float(loss[t]) - float(l64[0])) / abs(float(l64[0])))
        assert float(acc[t]) == float(a64[0])
    report(f'cfg4_T{T}', batched_vs_looped_grad_rel=max(eg), batched_vs_looped_loss_rel=max(el), vs_oracle_grad_rel=max(eo),
           vs_oracle_loss_rel=max(lo))
    assert max(el) < 1e-6 and max(eg) < 1e-5
    assert max(lo) < 1e-5 and max(eo) < 1e-4


def test_cfg3_anil_T32_batched_looped_oracle():
    """BASELINE config 3 (ANIL, 64-filter trunk on all 50 rows of a task, head-only inner loop, K = 1) at 32 tasks: the batched
    call vs the sum of single-task calls, and two single-task calls vs the fp64 oracle."""
    ways, shots, K, lr, T = 5, 5, 1, 0.5, 32
    base = R.convbase_spec(hidden=64, channels=3, max_pool=True)
    tf = OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic.ref_init_weights(R.param_shapes(base, '0.', False), 13).items())
    th = OrderedDict((k, torch.from_numpy(v)) for k, v in
                     synthetic.ref_init_weights(OrderedDict([('weight', (ways, 1600)), ('bias', (ways,))]), 17).items())
    theta = torch.cat([R.flatten_params(tf), R.flatten_params(th)]).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(ModelSpec.anil(ways))
    loss, acc, grad, _ = eng.meta_batch_anil(theta, d, l, shots, K, lr)
    torch.cuda.synchronize()
    gsum = torch.zeros_like(grad, dtype=torch.float64)
    el, per = [], {}
    for t in range(T):
        l1, a1, g1, _ = eng.meta_batch_anil(theta, d[t:t + 1], l[t:t + 1], shots, K, lr)
        torch.cuda.synchronize()
        gsum += g1.double()
        per[t] = (float(l1[0]), g1.double().cpu().numpy())
        el.append(abs(float(l1[0]) - float(loss[t])) / abs(float(loss[t])))
        assert float(a1[0]) == float(acc[t])
    e_sum = rel_err(grad.double().cpu().numpy(), gsum.cpu().numpy())
    eo, lo = [], []
    for t in (0, T - 1):
        l64, a64, gf, gh = R.anil_meta_batch(tf, th, base, 1600, [torch.from_numpy(data[t]).double()], [torch.from_numpy(labels[t])],
                                             K, shots, ways, lr, False)
        g64 = torch.cat([R.flatten_params(gf), R.flatten_params(gh)]).numpy()
        eo.append(rel_err(per[t][1], g64))
        lo.append(abs(per[t][0] - float(l64[0])) / abs(float(l64[0])))
    report('cfg3_anil_T32', batched_vs_looped_sum_grad_rel=e_sum, batched_vs_looped_loss_rel=max(el), vs_oracle_grad_rel=max(eo),
           vs_oracle_loss_rel=max(lo))
    assert max(el) < 1e-5 and e_sum < 1e-4
    assert max(lo) < 1e-4 and max(eo) < 2e-3
