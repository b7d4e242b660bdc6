"""Parity at the sizes bench.py times (BASELINE configs at full T), through the C ABI.

Why the checks are per step: with 5 inner steps at lr 0.5 the map theta_0 -> meta-gradient amplifies a 1e-7 perturbation to
1e-2..4e-1 (the reference's own fp32 run deviates from its fp64 run by 2e-1..4e-1 in the meta-gradient on these inputs --
measured with oracle/vision_ref.py, numbers in DESIGN.md section 7), so an end-to-end comparison cannot tell a correct
kernel from one with a 10 % error in the fifth Hessian-vector product.  The engine therefore dumps its per-step state
(mi_debug_set_trace: theta_k, g_k, the vector fed to every Hessian-vector product and its result) and the oracle is
TEACHER-FORCED: evaluated at the engine's own theta_k, in fp64 and in fp32 (the reference's own precision), for ALL 32 tasks.
Each step is then a single forward/backward (or one Hessian-vector product).

What the steps look like (profiles/r3/teacher_forced_cfg2_T32.md, all 160 steps): the median step agrees to 4e-7; 8-10 % of the
steps sit above 1e-4 and 2.5 % above 1e-3 (up to 6.5e-3) against EITHER leg -- and the two legs differ from each other by the
same amounts on other steps.  Cause, localised with tools/parity_localise.py and proved step by step here: the objective is only
piecewise smooth, one pass takes 1.9 million max-pool / ReLU decisions per task, and a handful of them per pass have an fp64
margin below fp32 rounding (|u| or the gap between the two largest u of a window < 1e-6).  ONE such decision moves a task's
gradient by up to ~5e-3 (it re-routes one element of a cotangent that carries ~1e-3 of the layer's gradient norm) and its
Hessian-vector product by as much (the tangent of the pooled value jumps to another position).  No fp32 arithmetic can reproduce
them: the reference's own fp32 leg flips a different subset.  So the bar has two parts:
  (1) raw, against each leg: median <= 1e-5 (HVP 1e-4), at most OUTLIER_SHARE of the steps above 1e-4, none above RAW_MAX;
  (2) NEAR-TIE ADJUSTED, for every step above 1e-5 (HVP 1e-4): tests/teacher_forced.py::explain_step searches the decisions whose
      fp64 margin is below TAU = 3e-6 for the assignment under which the fp64 arithmetic reproduces the engine; every step must then
      agree to ADJ_G (HVP ADJ_H).  A kernel error cannot pass (2): flipping near-tied decisions only adds the handful of discrete
      vectors those decisions control, it cannot imitate a dense error -- and systematic errors below 1e-5 are the business of the
      per-kernel tests (tests/test_gpu_tangent_kernels.py, 1e-6 at these sizes)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from exploring_meta_amd.engine import MetaEngine, ModelSpec
from exploring_meta_amd.utils import synthetic
from oracle import vision_ref as R
from gpu_utils import rel_err, report
import teacher_forced as TF

pytestmark = pytest.mark.gpu

# bars of the teacher-forced test (module docstring): raw maxima / outlier share against each leg, and the near-tie-adjusted maxima
# (RAW_MAX: the one-decision envelope, set to what is measured: cfg2's steps reached 6.5e-3 with the fp32 operand form
# and 4.4e-3 with the split-bf16 form, cfg3's tasks 1.3e-3, cfg4 at 32 tasks 5.7e-3.  Only cfg4's 256-task leg has drawn more -- one
# re-routed element worth 3.3e-2 of a one-shot step's gradient at lr 0.5 -- and carries its own bar, RAW_MAX_T256.  Both are draws,
# bounded here and explained decision by decision in part (2).  OUTLIER_SHARE is asserted against the fp64 leg: the reference's fp32
# leg is itself a draw of near-ties that depends on the host's thread count and BLAS, so its share is reported, not asserted.)
# FROZEN in round 5: RAW_MAX, RAW_MAX_T256, the one-decision envelope ENVELOPE of the batched-vs-looped comparison and the share
# bar against the reference's fp32 leg stay as they are; any further widening has to come with the offending decision (task, block, window,
# margin) in the failure output, which the assertions below now print.
RAW_MAX, RAW_MAX_T256, OUTLIER_SHARE, ADJ_G, ADJ_H = 3e-2, 1e-1, 0.15, 2e-5, 2e-4      # frozen r5
ENVELOPE, SHARE_FP32_LEG = 0.3, 0.3                                                     # frozen r5


def _decisions(res, worst=6):
    """The flipped near-tied decisions of a teacher-forced result list, as text for a failure message: (task, pass, block, window, kind,
    margin), largest margin first."""
    rows = []
    for r in res:
        flips = r['flips'] if r['flips'] and isinstance(r['flips'][0], list) else [r['flips']]
        for k, step in enumerate(flips):
            for fl in step:
                rows.append((abs(fl['margin']), f"task {r['t']} pass {k} block {fl['block']} at {tuple(fl['at'])} {fl['kind']} margin {fl['margin']:.2e}"))
    return [t for _, t in sorted(rows, reverse=True)[:worst]]


def _ref_theta(spec, seed=11):
    return OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic.ref_init_weights(R.param_shapes(spec), seed).items())


def _unflatten(flat, shapes):
    out, off = OrderedDict(), 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


def test_cfg2_T32_teacher_forced_per_step(conv_form_full):
    conv_form = conv_form_full
    """BASELINE config 2 exactly as benchmarked (32 tasks, 5-way 5-shot, K = 5, lr 0.5, second order, the bench's synthetic
    tasks and initial parameters): every inner-step gradient and every Hessian-vector product of ALL 32 tasks against the
    oracle at the engine's own theta_k (oracle legs in CPU worker processes); the theta recursion and the adjoint recursion
    themselves are checked exactly."""
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
    res = TF.teacher_forced_all(trace, data, labels, shots, ways, list(range(T)))
    legs = dict(g64=[], g32=[], h64=[], h32=[], q64=[], q32=[], gx=[], hx=[], qx=[])
    flips = []
    for r in res:
        t = r['t']
        report(f'cfg2_T32_teacher_forced[{conv_form}][task {t}]', grad_rel_vs_fp64=r['g64'], grad_rel_vs_ref_fp32=r['g32'], hvp_rel_vs_fp64=r['h64'],
               hvp_rel_vs_ref_fp32=r['h32'], query_grad_rel_vs_fp64=r['q64'][2], query_grad_rel_vs_ref_fp32=r['q32'][2],
               loss=float(loss[t]), loss_fp64=r['q64'][0], loss_ref_fp32=r['q32'][0])
        for k in ('g64', 'g32', 'h64', 'h32', 'gx', 'hx'):
            legs[k] += r[k]
        legs['qx'].append(r['qx'])
        flips += [fl for step in r['flips'] for fl in step]
        legs['q64'].append(r['q64'][2])
        legs['q32'].append(r['q32'][2])
        assert abs(float(loss[t]) - r['q64'][0]) <= 1e-5 * max(1.0, abs(r['q64'][0]))
        assert abs(float(loss[t]) - r['q32'][0]) <= 1e-5 * max(1.0, abs(r['q32'][0]))
        assert float(acc[t]) == r['q64'][1]
    for leg in ('64', '32'):      # (1) raw, against the reference arithmetic in fp64 and in the reference's own precision
        g, h, q = (np.array(legs[k + leg]) for k in 'ghq')
        report(f'cfg2_T32_teacher_forced[{conv_form}][all {T} tasks, leg fp{leg}]', grad_median=float(np.median(g)), grad_max=float(g.max()),
               grad_share_above_1e4=float((g > 1e-4).mean()), hvp_median=float(np.median(h)), hvp_max=float(h.max()),
               hvp_share_above_1e4=float((h > 1e-4).mean()), query_max=float(q.max()))
        share = OUTLIER_SHARE if leg == '64' else SHARE_FP32_LEG
        assert np.median(g) < 1e-5 and g.max() < RAW_MAX and (g > 1e-4).mean() <= share, (sorted(g)[-8:], _decisions(res))
        assert np.median(h) < 1e-4 and h.max() < RAW_MAX and (h > 1e-4).mean() <= share, (sorted(h)[-8:], _decisions(res))
        assert np.median(q) < 1e-5 and q.max() < RAW_MAX, sorted(q)[-4:]
    # (2) near-tie adjusted: every step, every task
    gx, hx, qx = (np.array(legs[k]) for k in ('gx', 'hx', 'qx'))
    margins = [abs(fl['margin']) for fl in flips]
    report(f'cfg2_T32_teacher_forced[{conv_form}][all {T} tasks, near-tie adjusted]', grad_max=float(gx.max()), hvp_max=float(hx.max()),
           query_max=float(qx.max()), flipped_decisions=len(flips), largest_flipped_margin=max(margins) if margins else 0.0)
    assert gx.max() < ADJ_G and qx.max() < ADJ_G, (sorted(gx)[-4:], sorted(qx)[-4:], _decisions(res))
    assert hx.max() < ADJ_H, (sorted(hx)[-4:], _decisions(res))
    assert all(m < TF.TAU for m in margins), _decisions(res)


@pytest.mark.parametrize('kernel', ['split_bf16_16x16', 'split_bf16_32x32'])
def test_cfg2_T32_batched_vs_one_task_at_a_time(kernel):
    """(Once per kernel of the default operand form, the SAME kernel in both runs -- conftest.CONV_FORMS: the engine's default picks the
    16x16x32 kernel by tiles per wave, i.e. for block 2 of the 32-task call only, and two kernels agree to fp32 rounding, not to the
    last bit: what this test pins is that ONE kernel's results do not depend on the launch geometry.)
    The batched launch against the same engine looped over single tasks (different launch geometry: one tile per wave,
    other partial-sum groupings).  Step 0 -- one forward/backward from identical parameters -- must agree to rounding for
    every task; after that the fp32 difference in partial-sum order (1e-7) is amplified by the chaotic inner loop exactly as
    the reference's own 1-thread-vs-8-thread difference is, so later steps are held to the teacher-forced
    test above and only reported here."""
    from conftest import apply_conv_form
    from exploring_meta_amd import _lib
    restore = apply_conv_form(_lib.load(), kernel)
    try:
        _batched_vs_one_task(kernel)
    finally:
        restore()


def _batched_vs_one_task(kernel):
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
    report(f'cfg2_T32_batched_vs_looped[{kernel}]', step0_grad_rel_max=max(e0), final_loss_rel_median=float(np.median(dl)),
           final_loss_rel_max=max(dl), acc_equal=accs_equal)
    assert max(e0) < 1e-5
    assert accs_equal >= T - 4 and float(np.median(dl)) < 5e-2


@pytest.mark.parametrize('T', [32, 256])
def test_cfg4_full_T_batched_looped_oracle(conv_form_full3, T):
    conv_form_full = conv_form_full3
    """(Every operand form at 32 tasks per GPU -- the benched size; the 256-task leg runs the default form only: its oracle legs take
    minutes.)  BASELINE config 4 (5-way 1-shot, one second-order step, 32 tasks per GPU; 256 = the whole meta-batch on one GPU) with
    the reference's initialisers and plateau-free inputs, the well-conditioned setting SURVEY.md 8c calibrated at <= 1e-4:
    per-task meta-gradients (from the trace) batched vs looped vs fp64 oracle."""
    if T == 256 and conv_form_full != 'split_bf16':
        pytest.skip('the 256-task leg runs the default operand form')
    if T == 256:
        # one kernel for the batched call and the one-task calls (the default picks by tiles per wave: the 16x16x32 kernel for block 2 of the
        # 256-task call only -- see test_cfg2_T32_batched_vs_one_task_at_a_time); restored by the fixture's own restore on exit
        from conftest import apply_conv_form
        from exploring_meta_amd import _lib
        apply_conv_form(_lib.load(), 'split_bf16_16x16')
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
    trace_all = {k: v.clone() for k, v in trace.items()}
    tr1 = eng.set_trace(1, K)
    eg, el, looped_traces = [], [], {}
    check = sorted(set(range(0, T, max(1, T // 16))) | {T - 1})
    for t in check:
        l1, a1, g1, _ = eng.meta_batch(theta, d[t:t + 1], l[t:t + 1], shots, K, lr)
        torch.cuda.synchronize()
        eg.append(rel_err(per_task[t].numpy(), g1.double().cpu().numpy()))
        el.append(abs(float(l1[0]) - float(loss[t])) / abs(float(loss[t])))
        assert float(a1[0]) == float(acc[t])
        if eg[-1] > 5e-3:                       # kept for the near-tie analysis of the ONE-TASK run below
            looped_traces[t] = {k: v.clone() for k, v in tr1.items()}
    eng.set_trace(0)
    eo, lo = [], []
    for t in sorted(set(range(0, T, max(1, T // 8))) | {T - 1}):
        best = []
        for dt in (torch.float64, torch.float32):
            thd = OrderedDict((k, v.to(dt)) for k, v in th64.items())
            lr_, ar_, gr_, _ = R.maml_meta_batch(thd, spec, [torch.from_numpy(data[t]).to(dt)], [torch.from_numpy(labels[t])], K, shots,
                                                 ways, lr, False)
            best.append(rel_err(per_task[t].numpy(), R.flatten_params(gr_).double().numpy()))
            if dt == torch.float64:
                lo.append(abs(float(loss[t]) - float(lr_[0])) / abs(float(lr_[0])))
                assert float(acc[t]) == float(ar_[0])
        eo.append(best)
    e64 = [b[0] for b in eo]
    ebest = [min(b) for b in eo]
    e2e_tasks = sorted(set(range(0, T, max(1, T // 8))) | {T - 1})
    # per step, teacher-forced, with the near-tie analysis of the cfg2 test (module docstring): 32 tasks (every 8th of 256)
    tf_tasks = sorted(set(range(0, T, max(1, T // 32))) | {T - 1})
    res = TF.teacher_forced_all(trace_all, data, labels, shots, ways, tf_tasks)
    raw = np.array([max(r['g64'][0], r['h64'][0], r['q64'][2]) for r in res])
    raw32 = np.array([max(r['g32'][0], r['h32'][0], r['q32'][2]) for r in res])
    adj_g = np.array([max(r['gx'][0], r['qx']) for r in res])
    adj_h = np.array([r['hx'][0] for r in res])
    margins = [abs(fl['margin']) for r in res for step in r['flips'] for fl in step]
    flipped = sorted(int(r['t']) for r in res if any(len(step) for step in r['flips']))
    report(f'cfg4_T{T}[{conv_form_full}]', end_to_end_tasks=e2e_tasks, tasks_with_near_tied_decisions=flipped, batched_vs_looped_grad_rel_median=float(np.median(eg)), batched_vs_looped_grad_rel_max=max(eg),
           batched_vs_looped_loss_rel=max(el), vs_fp64_grad_rel=e64, vs_fp64_or_ref_fp32_grad_rel=ebest, vs_fp64_loss_rel=lo,
           teacher_forced_tasks=len(res), teacher_forced_raw_max_vs_fp64=float(raw.max()), teacher_forced_raw_max_vs_fp32=float(raw32.max()),
           teacher_forced_adjusted_grad_max=float(adj_g.max()), teacher_forced_adjusted_hvp_max=float(adj_h.max()),
           flipped_decisions=len(margins), largest_flipped_margin=max(margins) if margins else 0.0)
    # One step at lr 0.5 from random weights overshoots (query loss 12..17): a task whose passes contain no near-tied pooling / ReLU
    # decision agrees to ~1e-6 end to end; one that does moves by 1e-4..2e-1 -- in the engine (either operand form of its hidden
    # convolutions: which way such a decision falls is a matter of the last bits, so the fp32 pipe and the split-bf16 form draw
    # different tasks) AND in the reference's own fp32 run (a task at 4.5e-2 from fp64 sits at 5e-6 from the reference's fp32 leg).
    # (The per-step analysis below does not flag every such task: a decision can also fall differently because the adapted
    # parameters the query pass starts from differ in their last bits -- teacher forcing removes exactly that.)  Hence, end to end:
    # at least two of the nine checked tasks agree with the nearer leg to 1e-5, the median to 1e-2, and none is off by more than the one-decision
    # envelope.  Per step: every checked task agrees with the fp64 arithmetic to ADJ_G / ADJ_H once the decisions with an fp64
    # margin below TAU (3e-6) are allowed to fall either way.
    # (batched against one-task calls: the launch geometry -- weight-gradient chunks, statistics partials -- follows the task count, so the
    # last bits differ and, with them, near-tied decisions: small ones (below 5e-3 of the task's gradient) in several tasks of the 256-task
    # leg, a larger one in at most a task or two of the seventeen checked -- the same one-decision envelope as against the oracle)
    assert max(el) < 1e-6 and np.median(eg) < 1e-5 and sum(e > 5e-3 for e in eg) <= 2 and max(eg) < ENVELOPE, eg
    # ... and a task above 5e-3 has to come with the decision that explains it: a near-tied (margin < TAU) pooling / ReLU decision that the
    # fp64 arithmetic needs flipped to reproduce the batched run or the one-task run of that task -- otherwise the difference is a
    # launch-geometry-dependent error, not a draw (round 4 advisor)
    by_task = {int(r['t']): r for r in res}
    for t, e in zip(check, eg):
        if e <= 5e-3:
            continue
        own = [by_task[t]] if t in by_task else TF.teacher_forced_all(trace_all, data, labels, shots, ways, [t])
        one = TF.teacher_forced_all(looped_traces[t], data[t:t + 1], labels[t:t + 1], shots, ways, [0])
        found = [fl for r in own + one for step in r['flips'] for fl in step]
        report(f'cfg4_T{T}[{conv_form_full}] batched-vs-looped outlier', task=int(t), grad_rel=float(e), decisions=_decisions(own + one))
        assert found and all(abs(fl['margin']) < TF.TAU for fl in found), \
            f'task {t}: batched vs one-task gradient differs by {e:.2e} without a near-tied decision to explain it: {_decisions(own + one)}'
    assert np.median(lo) < 1e-5 and max(lo) < 5e-3
    # (about half of the tasks hold such a decision: P(fewer than two clean ones among nine) is below 2 %)
    assert sum(e < 1e-5 for e in ebest) >= 2 and np.median(ebest) < 1e-2 and max(e64) < ENVELOPE, (e2e_tasks, ebest, flipped, _decisions(res))
    raw_max = RAW_MAX_T256 if T == 256 else RAW_MAX
    # (against the reference's fp32 leg the one-shot envelope RAW_MAX_T256 at either task count: there the draw is the REFERENCE's -- with
    # the fp16 operand form a task at 1.3e-6 from the fp64 leg sat at 3.8e-2 from the fp32 leg, whose own arithmetic re-routed an element)
    assert np.median(raw) < 1e-5 and raw.max() < raw_max and raw32.max() < RAW_MAX_T256
    assert adj_g.max() < ADJ_G and adj_h.max() < ADJ_H, (sorted(adj_g)[-4:], sorted(adj_h)[-4:], _decisions(res))
    assert all(m < TF.TAU for m in margins), _decisions(res)


def test_cfg3_anil_T32_batched_looped_oracle(conv_form_full):
    """(Every operand form of the 64-filter trunk's stride-1 convolutions.)  BASELINE config 3 (ANIL, 64-filter trunk on all 50 rows of a task, head-only inner loop, K = 1) at 32 tasks: the batched
    call vs the sum of single-task calls, and nine single-task calls vs the oracle in fp64 and fp32."""
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
    # 9 tasks against BOTH legs, with the near-tie analysis (tests/teacher_forced.py::anil_task): ANIL's trunk gradient is one
    # backward pass at theta -- not chaotic -- so a task either agrees to rounding or differs by the few near-tied decisions of
    # that pass, and then the fp64 arithmetic with those decisions flipped must reproduce it
    chk = sorted(set(range(0, T, 4)) | {T - 1})
    res = TF.anil_all(R.flatten_params(tf).float().numpy(), R.flatten_params(th).float().numpy(), {t: per[t][1] for t in chk}, data, labels,
                      shots, ways, K, lr, 64, chk)
    e64, e32, ex = (np.array([r[k] for r in res]) for k in ('e64', 'e32', 'ex'))
    lo = [abs(per[r['t']][0] - r['loss64']) / abs(r['loss64']) for r in res]
    margins = [abs(fl['margin']) for r in res for fl in r['flips']]
    report(f'cfg3_anil_T32[{conv_form_full}]', batched_vs_looped_sum_grad_rel=e_sum, batched_vs_looped_loss_rel=max(el), tasks_vs_oracle=len(res),
           vs_fp64_grad_rel=[float(x) for x in e64], vs_ref_fp32_grad_rel=[float(x) for x in e32], near_tie_adjusted_grad_rel=[float(x) for x in ex],
           vs_oracle_loss_rel=max(lo), flipped_decisions=len(margins), largest_flipped_margin=max(margins) if margins else 0.0)
    # batched vs looped: identical arithmetic per task (measured 1e-7)
    assert max(el) < 1e-5 and e_sum < 1e-4
    assert max(lo) < 1e-4 and e64.max() < RAW_MAX and e32.max() < RAW_MAX and np.median(np.minimum(e64, e32)) < 1e-4
    assert ex.max() < ADJ_G and all(m < TF.TAU for m in margins), (sorted(ex)[-4:], _decisions(res))
    assert all(float(acc[r['t']]) == r['acc64'] for r in res)


def test_hessian_vector_sweep_properties_at_cfg2_size():
    """Size-independent properties of the Hessian-vector sweep at BASELINE config 2's per-task size (25 images, 84x84, 32 filters),
    where the oracle would take minutes: with s(theta) = sum(logits * c) for a fixed cotangent c, H = d^2 s / dtheta^2 is
    symmetric and the sweep is linear in its direction -- <w, H v> = <v, H w>, H(a v + b w) = a H v + b H w, and J(a v + b w) =
    a J v + b J w for the logit tangents (mi_learner_hvp, the same tangent kernels the fused second-order path runs)."""
    ways, shots = 5, 5
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    theta = R.flatten_params(_ref_theta(spec, 42)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', [0, 1], ways, shots)
    x = torch.from_numpy(data[:, ::2]).float().cuda().contiguous()                  # the support halves: [2, 25, 3, 84, 84]
    T, n = x.shape[0], x.shape[1]
    eng = MetaEngine(mspec)
    g = torch.Generator(device='cuda').manual_seed(5)
    c = torch.randn(T, n, ways, device='cuda', generator=g) / n
    v = torch.randn(eng.param_count, device='cuda', generator=g) * 0.05
    w = torch.randn(eng.param_count, device='cuda', generator=g) * 0.05
    hv, jv = eng.learner_hvp(theta, x, c, v)
    hw, jw = eng.learner_hvp(theta, x, c, w)
    a, b = 0.7, -1.3
    hc, jc = eng.learner_hvp(theta, x, c, a * v + b * w)
    torch.cuda.synchronize()
    hv, hw, hc = hv[0].double(), hw[0].double(), hc[0].double()
    sym = abs(float(torch.dot(w.double(), hv) - torch.dot(v.double(), hw))) / max(abs(float(torch.dot(w.double(), hv))), 1e-30)
    lin = float((hc - (a * hv + b * hw)).norm() / hc.norm())
    linj = float((jc.double() - (a * jv.double() + b * jw.double())).norm() / jc.double().norm())
    report('hvp_properties_cfg2_size', symmetry_rel=sym, linearity_rel=lin, jvp_linearity_rel=linj)
    assert sym < 1e-4 and lin < 1e-5 and linj < 1e-5


@pytest.mark.parametrize('cfg', ['cfg2_T32', 'cfg4_T256'])
def test_last_arriver_fold_is_bit_identical_at_benched_size(cfg):
    """The last-arriver BatchNorm fold (csrc/finalize.h: write-through partials, drained, one relaxed agent-scope counter add per
    workgroup, the workgroup whose add came last folds with sc1 loads) at the sizes bench.py times -- hundreds of workgroups per
    launch across all 8 XCDs, where a visibility race would show, not the 2..7-task cases of test_fused_finalize_is_bit_identical:
    30 repeated fused-finalize calls against ONE call with separate bn_finalize launches.  Same fold order, so loss, accuracy,
    meta-gradient and the BatchNorm batch mean / variance of every block of every forward pass must be bit-identical every time
    (a stale partial would move a mean or a gradient in the last bits at least)."""
    if cfg == 'cfg2_T32':
        ways, shots, K, lr, T, seed = 5, 5, 5, 0.5, 32, 42
    else:
        ways, shots, K, lr, T, seed = 5, 1, 1, 0.5, 256, 11
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    theta = R.flatten_params(_ref_theta(spec, seed)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(mspec)

    def call():
        stats = eng.set_bn_export(T, K + 1)
        loss, acc, grad, _ = eng.meta_batch(theta, d, l, shots, K, lr)
        torch.cuda.synchronize()
        out = (loss.clone(), acc.clone(), grad.clone(), stats.clone())
        eng.set_bn_export(0)
        return out

    eng.set_fused_finalize(0)
    want = call()
    assert all(torch.isfinite(x).all() for x in want) and float(want[3].abs().sum()) > 0
    eng.set_fused_finalize(1)
    bad = []
    for rep in range(30):
        got = call()
        for name, a, b in zip(('loss', 'acc', 'grad', 'bn_stats'), got, want):
            if not torch.equal(a, b):
                bad.append((rep, name, int((a != b).sum())))
    report(f'last_arriver_fold_stress[{cfg}]', repeats=30, mismatches=len(bad))
    assert not bad, bad[:10]


@pytest.mark.parametrize('cfg', ['cfg2_T32', 'cfg4_T256'])
def test_fused_last_block_at_benched_size(cfg):
    """The one-launch tail of every pass (csrc/tail.hip, one workgroup per task) at the sizes bench.py times, against ONE call with the five
    separate launches per pass: 5 repeated fused calls are bit-identical to each other (fixed fold order, nothing depends on arrival order or
    placement), accuracy equals, and loss / meta-gradient / the BatchNorm batch statistics of every forward pass agree to 1e-6 at cfg4 (one
    step) -- at cfg2 the five chaotic steps amplify the last-bit difference of the BatchNorm-backward fold, so it is held to the
    bar two equally valid fp32 evaluation orders are held to elsewhere in this file (RAW_MAX)."""
    if cfg == 'cfg2_T32':
        ways, shots, K, lr, T, seed = 5, 5, 5, 0.5, 32, 42
    else:
        ways, shots, K, lr, T, seed = 5, 1, 1, 0.5, 256, 11
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    theta = R.flatten_params(_ref_theta(spec, seed)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(mspec)

    def call():
        stats = eng.set_bn_export(T, K + 1)
        loss, acc, grad, _ = eng.meta_batch(theta, d, l, shots, K, lr)
        torch.cuda.synchronize()
        out = (loss.clone(), acc.clone(), grad.clone(), stats.clone())
        eng.set_bn_export(0)
        return out

    eng.set_fused_last_block(0)
    want = call()
    eng.set_fused_last_block(1)
    first = call()
    for rep in range(4):
        got = call()
        for a, b in zip(got, first):
            assert torch.equal(a, b)
    exact = all(torch.equal(a, b) for a, b in zip(first, want))
    dl = float((first[0] - want[0]).abs().max())
    dg = float((first[2] - want[2]).norm() / want[2].norm())
    ds = float((first[3] - want[3]).abs().max() / want[3].abs().max())
    dacc = float((first[1] - want[1]).abs().max())
    report(f'fused_last_block_full_size[{cfg}]', bit_identical=bool(exact), loss_max_abs=dl, grad_rel=dg, bn_stats_rel=ds, acc_max_abs=dacc)
    if cfg == 'cfg4_T256':
        assert dl < 1e-6 and dg < 1e-6 and ds < 1e-6 and dacc == 0.0
    else:
        assert dl < RAW_MAX and dacc <= 0.04 + 1e-6


@pytest.mark.parametrize('cfg', ['cfg2_T32', 'cfg4_T256'])
def test_fused_tail_is_bit_identical_at_benched_size(cfg):
    """The one-launch pass tail (gram.hip advance_kernel) at the sizes bench.py times: its Gram statistics are formed by whichever
    workgroup of a task finishes last, from block-1 weights that other workgroups -- on other XCDs -- have just written (write-through
    stores, acknowledged, counted: the finalize.h protocol), and its folds run as slices that meet in LDS.  20 repeated fused calls
    against ONE call with the separate reduce / assemble / update / statistics launches: same arithmetic in the same order, so loss,
    accuracy, meta-gradient and the BatchNorm batch statistics of every block of every forward pass (block 1's come from the tail's
    statistics) must be bit-identical every time -- a stale weight or a lost arrival would move them."""
    if cfg == 'cfg2_T32':
        ways, shots, K, lr, T, seed = 5, 5, 5, 0.5, 32, 42
    else:
        ways, shots, K, lr, T, seed = 5, 1, 1, 0.5, 256, 11
    spec, mspec = R.mini_imagenet_spec(ways), ModelSpec.mini_imagenet(ways)
    theta = R.flatten_params(_ref_theta(spec, seed)).float().cuda().contiguous()
    data, labels = synthetic.make_meta_batch('min', list(range(T)), ways, shots)
    d, l = torch.from_numpy(data).cuda(), torch.from_numpy(labels).cuda()
    eng = MetaEngine(mspec)

    def call():
        stats = eng.set_bn_export(T, K + 1)
        loss, acc, grad, _ = eng.meta_batch(theta, d, l, shots, K, lr)
        torch.cuda.synchronize()
        out = (loss.clone(), acc.clone(), grad.clone(), stats.clone())
        eng.set_bn_export(0)
        return out

    eng.set_fused_tail(0)
    want = call()
    assert all(torch.isfinite(x).all() for x in want) and float(want[3].abs().sum()) > 0
    eng.set_fused_tail(1)
    bad = []
    for rep in range(20):
        got = call()
        for name, a, b in zip(('loss', 'acc', 'grad', 'bn_stats'), got, want):
            if not torch.equal(a, b):
                bad.append((rep, name, int((a != b).sum())))
    report(f'fused_tail_stress[{cfg}]', repeats=20, mismatches=len(bad))
    assert not bad, bad[:10]
