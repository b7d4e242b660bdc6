"""MAML-TRPO (BASELINE config 5) on the GPU against the fp64 oracle (oracle/rl_ref.py): policy density vs the reference's
own fixtures; trpo_update, meta surrogate loss / KL, its gradient, the Fisher-vector product and a whole
meta_optimize_trpo step vs the oracle on the same Particles2D replays."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from exploring_meta_amd import core_functions as cf
from oracle import rl_ref as RL
from helpers import hash_params
from gpu_utils import rel_err, report

pytestmark = pytest.mark.gpu

PARAMS = dict(inner_lr=0.1, max_path_length=25, adapt_steps=1, adapt_batch_size=6, meta_batch_size=4, outer_lr=0.3,
              backtrack_factor=0.5, ls_max_steps=15, max_kl=0.01, tau=1.0, gamma=0.99)


def _theta64():
    p = hash_params(RL.policy_param_shapes(), 19)
    p['sigma'] = torch.tensor([-0.3, 0.2], dtype=torch.float64)
    return p


def _policy(theta):
    pol = cf.DiagNormalPolicy(2, 2)
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(theta[k].float())
    return pol.cuda()


# BASELINE config 5 at full size: 20 tasks per meta-batch, 20 episodes x 100 steps per replay (rl/maml_trpo.py:21-33 defaults)
PARAMS_CFG5 = dict(PARAMS, max_path_length=100, adapt_batch_size=20, meta_batch_size=20)


def _replays(params=PARAMS):
    env = RL.Particles2D(seed=1)
    gen = torch.Generator().manual_seed(2)
    theta = _theta64()
    baseline = RL.LinearValue(2, 2)
    replays, olds = [], []
    for task in env.sample_tasks(params['meta_batch_size']):
        env.set_task(task)
        learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
        adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, baseline, params, gen, first_order=True)
        replays.append(rep)
        olds.append(OrderedDict((k, v.detach()) for k, v in adapted.items()))
    return theta, replays, olds


def test_policy_density_matches_reference(golden_small):
    pol = _policy(_theta64())
    st, ac = torch.from_numpy(golden_small['g5_states']).float().cuda(), torch.from_numpy(golden_small['g5_actions']).float().cuda()
    d = pol.density(st)
    assert np.allclose(d.loc.cpu().numpy(), golden_small['g5_policy_f64_loc'], atol=2e-6)
    assert np.allclose(d.scale.cpu().numpy(), golden_small['g5_policy_f64_scale'], rtol=1e-6)
    assert np.allclose(pol.log_prob(st, ac).cpu().numpy(), golden_small['g5_policy_f64_logp'], atol=5e-6)


def test_trpo_update_matches_oracle():
    theta, replays, olds = _replays()
    pol = _policy(theta)
    for t in range(len(replays)):
        new = cf.trpo_update(replays[t][0], pol, cf.LinearValue(2, 2), PARAMS['inner_lr'], PARAMS['gamma'], PARAMS['tau'])
        ref = torch.cat([v.reshape(-1) for v in olds[t].values()]).numpy()
        e = rel_err(new.flat().cpu().numpy() - pol.flat().cpu().numpy(), ref - torch.cat([v.reshape(-1) for v in theta.values()]).numpy())
        assert e < 1e-4, e


def test_surrogate_grad_fvp_match_oracle():
    theta, replays, olds = _replays()
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    loss, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), PARAMS)
    plist = list(p64.values())
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, plist, retain_graph=True)])
    Fvp = RL.hessian_vector_product(kl, plist)
    v = torch.randn(grad.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    fv = Fvp(v)

    pol = _policy(theta)
    old_pols = [_policy(o) for o in olds]
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), PARAMS)
    l32, k32, g32 = ctx.evaluate(pol.flat(), want_grad=True)
    f32 = ctx.fvp(pol.flat(), v.float().cuda())
    torch.cuda.synchronize()
    eg, ef = rel_err(g32.cpu().numpy(), grad.numpy()), rel_err(f32.cpu().numpy(), fv.detach().numpy())
    report('trpo_surrogate', loss=float(l32), loss_ref=float(loss), kl=float(k32), kl_ref=float(kl), grad_rel=eg, fvp_rel=ef)
    assert abs(float(l32) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert abs(float(k32) - float(kl)) < 1e-6          # both ~0: the adapted policy equals the stored old policy
    assert eg < 1e-4 and ef < 1e-3
    # a displaced candidate (line-search evaluation): values only
    cand = OrderedDict((k, (v.detach() + 0.01 * torch.sin(torch.arange(v.numel(), dtype=torch.float64)).view_as(v)).requires_grad_(True))
                       for k, v in theta.items())
    l2, k2 = RL.meta_surrogate_loss(replays, olds, cand, RL.LinearValue(2, 2), PARAMS)
    th2 = torch.cat([v.detach().reshape(-1) for v in cand.values()]).float().cuda()
    l2g, k2g, _ = ctx.evaluate(th2)
    assert abs(float(l2g) - float(l2)) < 2e-5 * max(1.0, abs(float(l2))) and abs(float(k2g) - float(k2)) < 1e-4 * max(float(k2), 1e-3)


def test_meta_optimize_trpo_matches_oracle():
    theta, replays, olds = _replays()
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    ref = RL.meta_optimize_trpo(PARAMS, p64, RL.LinearValue(2, 2), replays, olds)
    pol = _policy(theta)
    out = cf.meta_optimize_trpo(PARAMS, pol, cf.LinearValue(2, 2), replays, [_policy(o) for o in olds])
    es = rel_err(out['step'].cpu().numpy(), ref['step'].numpy())
    et = rel_err(pol.flat().cpu().numpy(), torch.cat([v.detach().reshape(-1) for v in p64.values()]).numpy())
    report('meta_optimize_trpo', step_rel=es, theta_rel=et, accepted=out['accepted'], accepted_ref=ref['accepted'])
    assert out['accepted'] == ref['accepted']
    assert es < 5e-3 and et < 1e-4


def test_cfg5_full_size_matches_oracle(golden_rl):
    """BASELINE config 5 as benchmarked -- 20 tasks x 2000-row replays -- against oracle/rl_ref.py: surrogate loss / KL, its
    gradient, a Fisher-vector product and the whole meta_optimize_trpo step (same accepted line-search index) -- AND against the
    records of the REFERENCE's own meta_surrogate_loss / meta_optimize_trpo (rl.py:409-473) executed on the same replays
    (tests/golden/golden_rl.npz, case cfg5)."""
    theta, replays, olds = _replays(PARAMS_CFG5)
    import rl_cases
    assert rl_cases.CASES['cfg5']['params'] == PARAMS_CFG5
    chk = rl_cases.input_checksums(replays)
    assert np.allclose(chk, golden_rl['rl_cfg5_f64_input_checksums'], rtol=1e-10, atol=1e-9), 'replays differ from the recorded ones'
    assert len(replays) == 20 and replays[0][0]['states'].shape[0] == 2000
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    loss, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), PARAMS_CFG5)
    plist = list(p64.values())
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, plist, retain_graph=True)])
    v = torch.randn(grad.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    fv = RL.hessian_vector_product(kl, plist)(v)
    pol = _policy(theta)
    old_pols = [_policy(o) for o in olds]
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), PARAMS_CFG5)
    l32, k32, g32 = ctx.evaluate(pol.flat(), want_grad=True)
    f32 = ctx.fvp(pol.flat(), v.float().cuda())
    f_first = ctx.fvp(pol.flat(), torch.from_numpy(golden_rl['rl_cfg5_f64_surr_grad']).float().cuda()).cpu().numpy()   # F g: the first CG product
    torch.cuda.synchronize()
    eg, ef = rel_err(g32.cpu().numpy(), grad.numpy()), rel_err(f32.cpu().numpy(), fv.detach().numpy())
    p64b = OrderedDict((k, v_.clone().requires_grad_(True)) for k, v_ in theta.items())
    ref = RL.meta_optimize_trpo(PARAMS_CFG5, p64b, RL.LinearValue(2, 2), replays, olds)
    out = cf.meta_optimize_trpo(PARAMS_CFG5, pol, cf.LinearValue(2, 2), replays, old_pols)
    es = rel_err(out['step'].cpu().numpy(), ref['step'].numpy())
    et = rel_err(pol.flat().cpu().numpy(), torch.cat([v_.detach().reshape(-1) for v_ in p64b.values()]).numpy())
    report('cfg5_full_size', loss=float(l32), loss_ref=float(loss.detach()), kl=float(k32), grad_rel=eg, fvp_rel=ef, step_rel=es,
           theta_rel=et, accepted=out['accepted'], accepted_ref=ref['accepted'])
    assert abs(float(l32) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach()))) and abs(float(k32)) < 1e-6
    assert eg < 1e-4 and ef < 1e-3
    assert out['accepted'] == ref['accepted'] and es < 5e-3 and et < 1e-4
    # the reference's own lines on these replays (fp64 record)
    G = lambda k: golden_rl['rl_cfg5_f64_' + k]
    egr = rel_err(g32.cpu().numpy(), G('surr_grad'))
    efr = rel_err(f_first, G('opt_fvp_first'))
    etr = rel_err(pol.flat().cpu().numpy(), G('opt_theta_new'))
    report('cfg5_full_size_vs_reference_record', loss_ref=float(G('surr_loss_kl')[0]), grad_rel=egr, fvp_first_rel=efr, theta_new_rel=etr,
           accepted_ref=int(G('opt_accepted')[0]), reference_fp32_rel_to_its_fp64=golden_rl['rl_cfg5_f32_rel_to_f64'].tolist())
    assert abs(float(l32) - G('surr_loss_kl')[0]) < 1e-5 and abs(float(k32) - G('surr_loss_kl')[1]) < 1e-6
    assert egr < 1e-4 and efr < 1e-3 and etr < 1e-4
    assert out['accepted'] == int(G('opt_accepted')[0])


@pytest.mark.parametrize('name', ['small_relu', 'two_steps'])
def test_trpo_path_matches_the_reference_records(golden_rl, name):
    """The HIP path against the records of the REFERENCE's rl.py executed on the stored replays (golden_rl.npz): trpo_a2c_loss /
    trpo_update (:346-374), fast_adapt_trpo's validation loss without a refit and success rate (:377-406), meta_surrogate_loss and
    its gradient (:441-473, :413-416), the first Fisher-vector product and the accepted step of meta_optimize_trpo (:409-438)."""
    import rl_cases
    case = rl_cases.load_case(golden_rl, name)
    params, theta, replays, olds = case['params'], case['theta'], case['replays'], case['olds']
    G = lambda k: golden_rl[f'rl_{name}_f64_' + k]
    pol = _policy(theta)
    flat0 = torch.cat([v.reshape(-1) for v in theta.values()]).numpy()
    # trpo_update on task 0's first support replay
    new = cf.trpo_update(replays[0][0], pol, cf.LinearValue(2, 2), params['inner_lr'], params['gamma'], params['tau'])
    e_up = rel_err(new.flat().cpu().numpy() - pol.flat().cpu().numpy(), G('adapted_theta') - flat0)
    l_in = cf.trpo_a2c_loss(replays[0][0], pol, cf.LinearValue(2, 2), params['gamma'], params['tau'])
    # fast_adapt_trpo with a runner that replays the stored episodes (as the generator drove the reference's)
    class Runner:
        def __init__(self, reps): self.reps, self.i = reps, 0
        def run(self, learner, episodes=None, render=False):
            self.i += 1
            return self.reps[self.i - 1]
    n_q = replays[0][-1]['states'].shape[0]
    q = dict(replays[0][-1], success=rl_cases.success_flags(n_q))
    adapted, vloss, _, rew, suc = cf.fast_adapt_trpo(Runner(replays[0][:-1] + [q]), _policy(theta), cf.LinearValue(2, 2), params, first_order=True)
    e_fa = rel_err(adapted.flat().cpu().numpy() - flat0, G('fa_theta') - flat0)
    # surrogate, gradient, first product, whole step
    old_pols = [_policy(o) for o in olds]
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), params)
    l32, k32, g32 = ctx.evaluate(pol.flat(), want_grad=True)
    f32 = ctx.fvp(pol.flat(), torch.from_numpy(G('surr_grad')).float().cuda())
    out = cf.meta_optimize_trpo(params, pol, cf.LinearValue(2, 2), replays, old_pols)
    torch.cuda.synchronize()
    eg, ef = rel_err(g32.cpu().numpy(), G('surr_grad')), rel_err(f32.cpu().numpy(), G('opt_fvp_first'))
    et = rel_err(pol.flat().cpu().numpy() - flat0, G('opt_theta_new') - flat0)
    report(f'trpo_vs_reference_record[{name}]', inner_loss=float(l_in), inner_loss_ref=float(G('inner_loss')[0]), update_rel=e_up,
           fast_adapt_rel=e_fa, valid_loss=float(vloss), valid_loss_ref=float(G('fa_valid_loss')[0]), grad_rel=eg, fvp_rel=ef,
           step_rel=et, accepted=out['accepted'])
    assert abs(float(l_in) - G('inner_loss')[0]) < 1e-5 and e_up < 1e-4 and e_fa < 1e-4
    assert abs(float(vloss) - G('fa_valid_loss')[0]) < 2e-5
    assert abs(rew - G('fa_reward_success')[0]) < 1e-4 * abs(G('fa_reward_success')[0]) and suc == G('fa_reward_success')[1]
    assert abs(float(l32) - G('surr_loss_kl')[0]) < 1e-5 and abs(float(k32) - G('surr_loss_kl')[1]) < 1e-6
    assert eg < 1e-4 and ef < 1e-3
    assert out['accepted'] == int(G('opt_accepted')[0]) and et < 5e-3


def test_cherry_style_replay_objects_take_the_same_path_as_dicts():
    """Replays as objects with cherry ExperienceReplay's accessors over CUDA tensors (what the reference's call sites hand over,
    rl.py:49-56) against the same replays as dicts: bit-identical step, same packed device batch (the object is read once)."""
    theta, replays, olds = _replays()

    class Episodes:
        def __init__(self, d): self._d = {k: v.float().cuda().contiguous() for k, v in d.items()}
        def state(self): return self._d['states']
        def action(self): return self._d['actions']
        def reward(self): return self._d['rewards']
        def done(self): return self._d['dones']
        def next_state(self): return self._d['next_states']
    objs = [[Episodes(r) for r in task] for task in replays]
    dicts = [[dict(e._d) for e in task] for task in objs]
    outs = []
    for reps in (objs, dicts):
        pol = _policy(theta)
        out = cf.meta_optimize_trpo(PARAMS, pol, cf.LinearValue(2, 2), reps, [_policy(o) for o in olds])
        outs.append((out['accepted'], out['step'].clone(), pol.flat().clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    from exploring_meta_amd.core_functions.rl import _as_replay, Replay
    assert type(_as_replay(objs[0][0])) is Replay and _as_replay(objs[0][0]) is _as_replay(objs[0][0])


def test_runner_and_fast_adapt_trpo_shapes():
    pol = _policy(_theta64())
    gen = torch.Generator(device='cuda').manual_seed(0)
    task = cf.Particles2DRunner([0.2, -0.3], PARAMS['max_path_length'], gen)
    learner, valid_loss, replay, rew, _ = cf.fast_adapt_trpo(task, pol, cf.LinearValue(2, 2), PARAMS, first_order=True)
    assert len(replay) == 2 and replay[0]['states'].shape[1] == 2 and replay[0]['dones'].sum().item() == PARAMS['adapt_batch_size']
    assert torch.isfinite(valid_loss) and rew < 0
    assert not torch.equal(learner.flat(), pol.flat())


# ---------------------------------------------------------------------------------------------- tanh policies (policies.py:32-37,70-126)
def _theta64_tanh():
    return _theta64()


def _policy_tanh(theta):
    pol = cf.DiagNormalPolicy(2, 2, activation='tanh')
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(theta[k].float())
    return pol.cuda()


def _replays_tanh(anil=False, params=PARAMS):
    env = RL.Particles2D(seed=1)
    gen = torch.Generator().manual_seed(2)
    theta = _theta64_tanh()
    baseline = RL.LinearValue(2, 2)
    replays, olds = [], []
    for task in env.sample_tasks(params['meta_batch_size']):
        env.set_task(task)
        learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
        adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, baseline, params, gen, first_order=True, activation=torch.tanh, anil=anil)
        replays.append(rep)
        olds.append(OrderedDict((k, v.detach()) for k, v in adapted.items()))
    return theta, replays, olds


def test_anil_policy_density_matches_reference(golden_small):
    """DiagNormalPolicyANIL (tanh body + head, sigma) against the reference's own module (fixture G5)."""
    raw = hash_params(RL.anil_policy_param_shapes(), 23)
    pol = cf.DiagNormalPolicyANIL(2, 2, 100)
    assert [k for k, _ in pol.named_parameters()] == list(raw.keys())        # sigma, body.0.*, body.2.*, head.*
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(raw[k].float())
    pol = pol.cuda()
    st, ac = torch.from_numpy(golden_small['g5_states']).float().cuda(), torch.from_numpy(golden_small['g5_actions']).float().cuda()
    assert np.allclose(pol.log_prob(st, ac).cpu().numpy(), golden_small['g5_anil_f64_bodyoff0_logp'], atol=5e-6)
    pol.turn_off_body_grads()
    assert np.allclose(pol.log_prob(st, ac).cpu().numpy(), golden_small['g5_anil_f64_bodyoff1_logp'], atol=5e-6)


def test_anil_trpo_update_moves_only_head_and_sigma():
    theta, replays, olds = _replays_tanh(anil=True)
    raw = OrderedDict()
    names = list(RL.anil_policy_param_shapes().keys())
    for k_anil, (k, v) in zip(names, theta.items()):
        raw[k_anil] = v
    pol = cf.DiagNormalPolicyANIL(2, 2, 100)
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(raw[k].float())
    pol = pol.cuda()
    pol.turn_off_body_grads()
    for t in range(len(replays)):
        new = cf.trpo_update(replays[t][0], pol, cf.LinearValue(2, 2), PARAMS['inner_lr'], PARAMS['gamma'], PARAMS['tau'], anil=True)
        ref = torch.cat([v.reshape(-1) for v in olds[t].values()]).numpy()
        base = torch.cat([v.reshape(-1) for v in theta.values()]).numpy()
        d, dref = new.flat().cpu().numpy() - pol.flat().cpu().numpy(), ref - base
        assert rel_err(d, dref) < 1e-4
        body = slice(2, 2 + 100 * 2 + 100 + 100 * 100 + 100)                 # W1, b1, W2, b2 in the engine's flat order
        assert np.all(d[body] == 0.0) and np.all(dref[body] == 0.0) and np.abs(d[:2]).max() > 0


def _anil_policy(theta):
    raw = OrderedDict((ka, v) for ka, (k, v) in zip(RL.anil_policy_param_shapes().keys(), theta.items()))
    pol = cf.DiagNormalPolicyANIL(2, 2, 100)
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(raw[k].float())
    return pol.cuda()


@pytest.mark.parametrize('act', ['tanh', 'relu'])
def test_general_kl_hvp_matches_autograd(act):
    """ANIL-TRPO (rl/anil_trpo.py:129, rl.py:409-473 with anil=True): the old policies were adapted head-only, the surrogate
    re-adapts every parameter, so new != old and trpo.hessian_vector_product(kl) is the exact Hessian of the mean KL --
    J^T Hess KL J v - lr T[v, grad KL] with the third derivative of the inner loss (mi_trpo_kl_prepare / mi_trpo_fvp_general).
    Checked against double-backward autograd of the oracle's meta_surrogate_loss: the KL value, its gradient and Hessian-vector
    products along a random and a structured direction."""
    activation = torch.tanh if act == 'tanh' else torch.relu
    if act == 'tanh':
        theta, replays, olds = _replays_tanh(anil=True)
    else:                                        # ReLU body with head-only old policies: phi'' = 0, the other third-order paths remain
        env, gen, theta, baseline = RL.Particles2D(seed=1), torch.Generator().manual_seed(2), _theta64(), RL.LinearValue(2, 2)
        replays, olds = [], []
        for task in env.sample_tasks(PARAMS['meta_batch_size']):
            env.set_task(task)
            learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
            adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, baseline, PARAMS, gen, first_order=True, anil=True)
            replays.append(rep)
            olds.append(OrderedDict((k, v.detach()) for k, v in adapted.items()))
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    loss, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), PARAMS, activation=activation)
    plist = list(p64.values())
    assert float(kl) > 1e-6                                                # new != old: this is not the Fisher case
    gkl = torch.cat([g.reshape(-1) for g in torch.autograd.grad(kl, plist, retain_graph=True)])
    Hvp = RL.hessian_vector_product(kl, plist)
    gen = torch.Generator().manual_seed(5)
    vs = [torch.randn(gkl.shape, generator=gen, dtype=torch.float64), gkl / gkl.norm()]
    refs = [Hvp(v).detach() for v in vs]

    pol = _policy_tanh(theta) if act == 'tanh' else _policy(theta)
    old_pols = [(_policy_tanh(o) if act == 'tanh' else _policy(o)) for o in olds]
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), PARAMS)
    th = pol.flat()
    l32, k32, _ = ctx.evaluate(th, want_grad=True)
    g32 = ctx.engine.kl_prepare(th, ctx.sup, ctx.qry, ctx.old_loc, ctx.old_scale, ctx.inner_lr, want_grad=True)
    ctx.general = True
    errs = [rel_err(ctx.fvp(th, v.float().cuda()).cpu().numpy(), r.numpy()) for v, r in zip(vs, refs)]
    # the Fisher form (valid only at new == old) must NOT reproduce these products: the third-order and cotangent terms matter
    ctx.general = False
    fisher_err = rel_err(ctx.fvp(th, vs[0].float().cuda()).cpu().numpy(), refs[0].numpy())
    torch.cuda.synchronize()
    eg = rel_err(g32.cpu().numpy(), gkl.detach().numpy())
    report(f'general_kl_hvp[{act}]', kl=float(k32), kl_ref=float(kl), kl_grad_rel=eg, hvp_rel=errs, fisher_form_rel=fisher_err)
    assert abs(float(k32) - float(kl)) < 1e-5 * max(float(kl), 1e-3)
    assert eg < 1e-4
    assert max(errs) < 1e-3
    assert fisher_err > 10 * max(errs)


def test_meta_optimize_anil_trpo_matches_oracle():
    """One whole ANIL-TRPO meta-optimisation (conjugate gradient on the exact KL Hessian, line search) against the oracle."""
    # inner_lr as in rl/anil_trpo.py:22.  (At the larger MAML-TRPO step the re-adapted policies sit far from the stored ones, the
    # exact KL Hessian is indefinite, and the reference's own step is NaN: sqrt of a negative s.Hs -- no update either way.)
    P = dict(PARAMS, inner_lr=0.01)
    theta, replays, olds = _replays_tanh(anil=True, params=P)
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    ref = RL.meta_optimize_trpo(P, p64, RL.LinearValue(2, 2), replays, olds, activation=torch.tanh)
    assert ref['accepted'] is not None
    pol = _anil_policy(theta)
    old_pols = [_anil_policy(o) for o in olds]
    out = cf.meta_optimize_trpo(P, pol, cf.LinearValue(2, 2), replays, old_pols, anil=True)
    es = rel_err(out['step'].cpu().numpy(), ref['step'].numpy())
    et = rel_err(pol.flat().cpu().numpy(), torch.cat([v.detach().reshape(-1) for v in p64.values()]).numpy())
    eg = rel_err(out['grad'].cpu().numpy(), ref['grad'].numpy())
    # the same conjugate-gradient recurrences in fp64 on the ENGINE's products: separates the product's accuracy from the
    # sensitivity of ten CG iterations on a nearly singular system (damping 1e-5) to fp32 rounding in A p
    ctx = out['context']
    th0 = torch.cat([v.reshape(-1) for v in theta.values()]).float().cuda()
    ctx.evaluate(th0, want_grad=True)
    ctx.prepare_general_kl(th0)
    step64 = RL.conjugate_gradient(lambda v: ctx.fvp(th0, v.float().cuda()).double().cpu(), ref['grad'])
    shs = 0.5 * torch.dot(step64, ctx.fvp(th0, step64.float().cuda()).double().cpu())
    step64 = step64 / torch.sqrt(shs / P['max_kl'])
    ec = rel_err(step64.numpy(), ref['step'].numpy())
    # ... and the oracle's own sensitivity: its fp64 products and right-hand side rounded to fp32 (what the reference's fp32 run does)
    r32 = lambda x: x.float().double()
    s32 = RL.conjugate_gradient(lambda v: r32(ref['fvp'](r32(v))), r32(ref['grad']))
    s32 = s32 / torch.sqrt(0.5 * torch.dot(s32, ref['fvp'](s32)) / P['max_kl'])
    e32 = rel_err(s32.numpy(), ref['step'].numpy())
    report('meta_optimize_anil_trpo', step_rel=es, theta_rel=et, grad_rel=eg, step_rel_fp64_cg_on_engine_products=ec,
           step_rel_oracle_with_fp32_rounded_products=e32, accepted=out['accepted'], accepted_ref=ref['accepted'])
    assert out['accepted'] == ref['accepted']
    assert eg < 1e-5
    # Ten CG iterations on a nearly singular system (damping 1e-5) amplify a 6e-8 (fp32 rounding) perturbation of the right-hand
    # side into e32 = O(0.04) of the step direction -- in the reference's own arithmetic.  With the exact right-hand side the
    # engine's products reproduce the oracle's step as well as the oracle's fp32-rounded products do (ec ~ e32); the whole
    # pipeline adds the engine's 1.6e-7 gradient error, amplified by the same factor.
    assert ec <= max(2e-2, 4 * e32)            # (a different but equally valid fp32 summation order inside the products gave 2.0 x e32)
    amp = e32 / 6e-8
    assert es <= max(2e-2, 4 * e32 + 2 * amp * eg)
    tn = float(torch.cat([v.detach().reshape(-1) for v in p64.values()]).norm())
    assert et <= max(1e-3, es * float(ref['step'].norm()) * P['outer_lr'] / tn * 2)


def test_anil_trpo_matches_the_reference_record(golden_rl):
    """ANIL-TRPO (rl/anil_trpo.py; rl.py:381-382,395-396,409-473 with anil=True) against the record of the REFERENCE's own lines on the stored
    replays (golden_rl.npz, case anil_tanh): the head-only inner update of fast_adapt_trpo, the surrogate loss / KL at parameters where the
    re-adapted policies differ from the stored ones, its gradient, and the first product of the exact KL Hessian (the third-derivative
    terms included).  The conjugate-gradient STEP is not compared: the record holds the reference's own fp32 run beside its fp64 run, and
    their steps differ by 0.88 of the step's norm (rl_anil_tanh_f32_rel_to_f64[3]) -- ten iterations on a nearly singular, indefinite system."""
    import rl_cases
    case = rl_cases.load_case(golden_rl, 'anil_tanh')
    params, theta, replays, olds = case['params'], case['theta'], case['replays'], case['olds']
    G = lambda k: golden_rl['rl_anil_tanh_f64_' + k]
    flat0 = torch.cat([v.reshape(-1) for v in theta.values()]).numpy()
    pol = _anil_policy(theta)
    pol.turn_off_body_grads()
    new = cf.trpo_update(replays[0][0], pol, cf.LinearValue(2, 2), params['inner_lr'], params['gamma'], params['tau'], anil=True)
    pol.turn_on_body_grads()
    e_fa = rel_err(new.flat().cpu().numpy() - flat0, G('fa_theta') - flat0)
    old_pols = [_anil_policy(o) for o in olds]
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), params)
    th = pol.flat()
    l32, k32, g32 = ctx.evaluate(th, want_grad=True)
    ctx.prepare_general_kl(th)
    f32 = ctx.fvp(th, torch.from_numpy(G('surr_grad')).float().cuda())
    torch.cuda.synchronize()
    eg, ef = rel_err(g32.cpu().numpy(), G('surr_grad')), rel_err(f32.cpu().numpy(), G('opt_fvp_first'))
    report('anil_trpo_vs_reference_record', fast_adapt_rel=e_fa, loss=float(l32), loss_ref=float(G('surr_loss_kl')[0]), kl=float(k32),
           kl_ref=float(G('surr_loss_kl')[1]), grad_rel=eg, fvp_first_rel=ef, reference_fp32_rel_to_its_fp64=golden_rl['rl_anil_tanh_f32_rel_to_f64'].tolist())
    # (the update is lr 0.01 x a gradient of norm ~0.1 on parameters of size 0.3: the DIFFERENCE carries fp32's 3e-8 parameter rounding as ~1e-4 of its own norm)
    assert e_fa < 5e-4
    assert abs(float(l32) - G('surr_loss_kl')[0]) < 1e-6 and abs(float(k32) - G('surr_loss_kl')[1]) < 1e-5 * max(G('surr_loss_kl')[1], 1e-3)
    assert eg < 1e-4 and ef < 1e-3


def test_tanh_surrogate_grad_fvp_match_oracle():
    """MAML-TRPO with DiagNormalPolicy(activation='tanh'): the tanh curvature term of the inner-loss HVP is exercised by the
    surrogate gradient (I - lr H) grad S and by both H products inside the Fisher-vector product."""
    theta, replays, olds = _replays_tanh()
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    loss, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), PARAMS, activation=torch.tanh)
    plist = list(p64.values())
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, plist, retain_graph=True)])
    Fvp = RL.hessian_vector_product(kl, plist)
    v = torch.randn(grad.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    fv = Fvp(v)
    pol = _policy_tanh(theta)
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, [_policy_tanh(o) for o in olds], pol, cf.LinearValue(2, 2), PARAMS)
    l32, k32, g32 = ctx.evaluate(pol.flat(), want_grad=True)
    f32 = ctx.fvp(pol.flat(), v.float().cuda())
    eg, ef = rel_err(g32.cpu().numpy(), grad.numpy()), rel_err(f32.cpu().numpy(), fv.detach().numpy())
    report('trpo_surrogate_tanh', loss=float(l32), loss_ref=float(loss), kl=float(k32), kl_ref=float(kl), grad_rel=eg, fvp_rel=ef)
    assert abs(float(l32) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert abs(float(k32) - float(kl)) < 1e-6
    assert eg < 1e-4 and ef < 1e-3
    # a larger inner step makes the curvature term matter: first-order-only HVP would be far off
    big = dict(PARAMS, inner_lr=1.0)
    loss2, _ = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), big, activation=torch.tanh)
    grad2 = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss2, plist)])
    ctx2 = _SurrogateContext(replays, [_policy_tanh(o) for o in olds], pol, cf.LinearValue(2, 2), big)
    _, _, g2 = ctx2.evaluate(pol.flat(), want_grad=True)
    assert rel_err(g2.cpu().numpy(), grad2.numpy()) < 2e-4


def test_tanh_meta_optimize_trpo_matches_oracle():
    theta, replays, olds = _replays_tanh()
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    ref = RL.meta_optimize_trpo(PARAMS, p64, RL.LinearValue(2, 2), replays, olds, activation=torch.tanh)
    pol = _policy_tanh(theta)
    out = cf.meta_optimize_trpo(PARAMS, pol, cf.LinearValue(2, 2), replays, [_policy_tanh(o) for o in olds])
    es = rel_err(out['step'].cpu().numpy(), ref['step'].numpy())
    et = rel_err(pol.flat().cpu().numpy(), torch.cat([v.detach().reshape(-1) for v in p64.values()]).numpy())
    report('meta_optimize_trpo_tanh', step_rel=es, theta_rel=et, accepted=out['accepted'], accepted_ref=ref['accepted'])
    assert out['accepted'] == ref['accepted']
    # Ten CG iterations on the tanh policy's Fisher matrix are ill-conditioned: the ORACLE run in fp32 already sits 9.2e-3
    # (step) / 6.2e-4 (theta) away from its own fp64 run on these replays (ReLU: 4.1e-5 / 2.3e-6), while each engine
    # Fisher-vector product agrees with fp64 to 2e-7 (test above; measured here: 1.3e-2 / 8.6e-4 with fp64 CG recurrences).
    # Bar: 2x the reference's own fp32 deviation.
    assert es < 2e-2 and et < 1.3e-3


# ---------------------------------------------------------------------------------------------- VPG / PPO inner loops (rl.py:209-337)
def _stack_batches(eps_per_task_per_batch, advs, dev='cuda'):
    """eps[nb][t] replay dicts + advantages -> {states [NB,T,B,S], actions, adv [NB,T,B], count [NB,T]} padded to a common B."""
    NB, T = len(eps_per_task_per_batch), len(eps_per_task_per_batch[0])
    B = max(int(e['states'].shape[0]) for row in eps_per_task_per_batch for e in row)
    st, ac = torch.zeros(NB, T, B, 2), torch.zeros(NB, T, B, 2)
    ad, cnt, dn = torch.zeros(NB, T, B), torch.zeros(NB, T, dtype=torch.int32), torch.zeros(NB, T, B)
    for b in range(NB):
        for t in range(T):
            e, a = eps_per_task_per_batch[b][t], advs[b][t]
            n = int(e['states'].shape[0])
            st[b, t, :n], ac[b, t, :n], ad[b, t, :n], cnt[b, t] = e['states'].float(), e['actions'].float(), a.reshape(-1).float(), n
            dn[b, t, :n] = e['dones'].reshape(-1).float()
    return dict(states=st.to(dev), actions=ac.to(dev), adv=ad.to(dev), count=cnt.to(dev), done=dn.to(dev)), B


@pytest.mark.parametrize('algo,act,anil,first_order,lr', [('vpg', 'relu', False, False, 0.05), ('vpg', 'tanh', True, False, 0.05),
                                                            ('vpg', 'relu', False, True, 0.05), ('ppo', 'relu', False, False, 0.05),
                                                            ('ppo', 'tanh', False, False, 0.05), ('ppo', 'tanh', True, False, 0.05),
                                                            ('ppo', 'relu', False, False, 2.0),    # large steps: the clip is active
                                                            # vpg_a2c_loss(dice=True), rl.py:219-226: the episode recurrences of
                                                            # weighted_cumsum couple the samples (gradient M^T a, HVP M^T diag(a) M)
                                                            ('dice', 'relu', False, False, 0.05), ('dice', 'tanh', True, False, 0.05),
                                                            ('dice', 'relu', False, True, 0.05), ('dice', 'tanh', False, False, 0.5)])
def test_policy_meta_batch_vpg_ppo(algo, act, anil, first_order, lr):
    """mi_policy_meta_batch against the autograd restatement of fast_adapt_vpg / fast_adapt_ppo on the same replays: validation
    loss, adapted parameters and the (second-order) meta-gradient `av_loss.backward()` leaves behind (rl/maml_ppo.py:129)."""
    T, adapt_steps = 3, 2
    P = dict(tau=1.0, gamma=0.99, inner_lr=lr, ppo_epochs=3, ppo_clip_ratio=0.1 if lr < 1 else 0.02)
    activation = torch.relu if act == 'relu' else torch.tanh
    theta = _theta64()
    env = RL.Particles2D(seed=3)
    gen = torch.Generator().manual_seed(4)
    sup, qry = [[None] * T for _ in range(adapt_steps)], [None] * T
    for t, task in enumerate(env.sample_tasks(T)):
        env.set_task(task)
        for b in range(adapt_steps):
            sup[b][t] = RL.collect_episodes(env, theta, 5, 12 + 3 * b, gen, activation=activation)
        qry[t] = RL.collect_episodes(env, theta, 5, 15, gen, activation=activation)
    # oracle: per-task loss, adapted parameters, gradient; advantages from the (deterministic) baseline fits it performs
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    losses, thetas, gsum = [], [], torch.zeros(sum(v.numel() for v in theta.values()), dtype=torch.float64)
    for t in range(T):
        s_t = [sup[b][t] for b in range(adapt_steps)]
        if algo in ('vpg', 'dice'):
            loss, pk = RL.replay_vpg(leaves, s_t, qry[t], P, RL.LinearValue(2, 2), first_order=first_order, activation=activation, anil=anil,
                                     dice=algo == 'dice')
        else:
            loss, pk = RL.replay_ppo(leaves, s_t, qry[t], P, RL.LinearValue(2, 2), activation=activation, anil=anil)
        g = torch.autograd.grad(loss, list(leaves.values()))
        gsum += torch.cat([x.reshape(-1) for x in g])
        losses.append(float(loss))
        thetas.append(torch.cat([v.detach().reshape(-1) for v in pk.values()]))
    # the same advantages for the engine
    def adv_of(ep):
        a = RL.compute_advantages(RL.LinearValue(2, 2), P['tau'], P['gamma'], ep)
        return (RL.normalize(a) if algo == 'ppo' else a).detach()
    sup_b, B1 = _stack_batches(sup, [[adv_of(e) for e in row] for row in sup])
    qry_b, B2 = _stack_batches([qry], [[adv_of(e) for e in qry]])
    B = max(B1, B2)
    def padB(d):
        out = {}
        for k, v in d.items():
            if k == 'count':
                out[k] = v
            else:
                shp = list(v.shape); ax = 2
                padn = B - shp[ax]
                out[k] = torch.nn.functional.pad(v, ((0, 0, 0, padn) if v.dim() == 4 else (0, padn))).contiguous()
        return out
    sup_b, qry_b = padB(sup_b), {k: v[0] for k, v in padB(qry_b).items()}
    pol = (_policy_tanh if act == 'tanh' else _policy)(theta)
    eng = pol.engine()
    epochs = P['ppo_epochs'] if algo == 'ppo' else 1
    step_batch = [b for b in range(adapt_steps) for _ in range(epochs)]
    kind = {'ppo': 'ppo', 'vpg': 'a2c', 'dice': 'dice'}[algo]
    loss, th_out, grad = eng.meta_batch(pol.flat(), sup_b, qry_b, step_batch, P['inner_lr'], loss=kind,
                                        clip=P['ppo_clip_ratio'], head_only=anil, first_order=first_order)
    torch.cuda.synchronize()
    e_th = max(rel_err(th_out[t].cpu().numpy() - pol.flat().cpu().numpy(),
                       thetas[t].numpy() - torch.cat([v.reshape(-1) for v in theta.values()]).numpy()) for t in range(T))
    e_g = rel_err(grad.cpu().numpy(), gsum.numpy())
    e_l = max(abs(float(loss[t]) - losses[t]) for t in range(T))
    report(f'policy_meta[{algo},{act},anil={anil},fo={first_order},lr={lr}]', loss_abs=e_l, theta_step_rel=e_th, grad_rel=e_g)
    assert e_l < 2e-6 * max(1.0, max(abs(x) for x in losses))
    assert e_th < 1e-4 and e_g < 2e-4
    if anil:                                                             # the body did not move during adaptation
        body = slice(2, 2 + 100 * 2 + 100 + 100 * 100 + 100)
        assert torch.equal(th_out[:, body], pol.flat()[body].expand(T, -1))
    # evaluation-only call: same loss, no gradient
    l2, _, g2 = eng.meta_batch(pol.flat(), sup_b, qry_b, step_batch, P['inner_lr'], loss=kind,
                               clip=P['ppo_clip_ratio'], head_only=anil, first_order=first_order, with_grad=False)
    assert g2 is None and torch.allclose(l2, loss, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('act', ['relu', 'tanh'])
def test_trpo_two_adapt_steps_match_oracle(act):
    """params['adapt_steps'] = 2: surrogate, its gradient (adjoint through both updates) and the Fisher-vector product
    J^T F J v (tangent through both updates, Fisher at the query, adjoint back) -- mi_trpo_surrogate_steps / mi_trpo_fvp_steps."""
    P2 = dict(PARAMS, adapt_steps=2)
    activation = torch.relu if act == 'relu' else torch.tanh
    env = RL.Particles2D(seed=1)
    gen = torch.Generator().manual_seed(2)
    theta = _theta64()
    replays, olds = [], []
    for task in env.sample_tasks(P2['meta_batch_size']):
        env.set_task(task)
        learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
        adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, RL.LinearValue(2, 2), P2, gen, first_order=True, activation=activation)
        assert len(rep) == 3
        replays.append(rep)
        olds.append(OrderedDict((k, v.detach()) for k, v in adapted.items()))
    p64 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
    loss, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), P2, activation=activation)
    plist = list(p64.values())
    grad = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, plist, retain_graph=True)])
    Fvp = RL.hessian_vector_product(kl, plist)
    v = torch.randn(grad.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    fv = Fvp(v)
    mk = _policy if act == 'relu' else _policy_tanh
    pol = mk(theta)
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    ctx = _SurrogateContext(replays, [mk(o) for o in olds], pol, cf.LinearValue(2, 2), P2)
    assert ctx.steps == 2
    l32, k32, g32 = ctx.evaluate(pol.flat(), want_grad=True)
    f32 = ctx.fvp(pol.flat(), v.float().cuda())
    eg, ef = rel_err(g32.cpu().numpy(), grad.detach().numpy()), rel_err(f32.cpu().numpy(), fv.detach().numpy())
    report(f'trpo_two_steps[{act}]', loss=float(l32), loss_ref=float(loss.detach()), kl=float(k32), kl_ref=float(kl.detach()), grad_rel=eg, fvp_rel=ef)
    assert abs(float(l32) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach())))
    assert abs(float(k32) - float(kl.detach())) < 1e-6
    assert eg < 1e-4 and ef < 1e-3
    # whole meta-optimisation step
    ref = RL.meta_optimize_trpo(P2, p64, RL.LinearValue(2, 2), replays, olds, activation=activation)
    out = cf.meta_optimize_trpo(P2, pol, cf.LinearValue(2, 2), replays, [mk(o) for o in olds])
    assert out['accepted'] == ref['accepted']
    es = rel_err(out['step'].cpu().numpy(), ref['step'].numpy())
    report(f'trpo_two_steps_meta_optimize[{act}]', step_rel=es, accepted=out['accepted'])
    assert es < (5e-3 if act == 'relu' else 3e-2)


# ---------------------------------------------------------------------------------------------- advantages on the GPU
def _random_replay(seed, n, S, ep_len, truncated=False):
    g = torch.Generator().manual_seed(seed)
    dones = torch.zeros(n, 1, dtype=torch.float64)
    dones[ep_len - 1::ep_len] = 1.0
    if n > 9:
        dones[7] = 1.0                               # one early termination
    if truncated:
        dones[-1] = 0.0                              # the last episode is cut by the replay's end, not by a done
    f32 = lambda t: t.float().double()               # values the kernel's fp32 inputs represent exactly
    return dict(states=f32(torch.randn(n, S, generator=g, dtype=torch.float64)), actions=torch.randn(n, 2, generator=g, dtype=torch.float64),
                rewards=f32(-torch.rand(n, 1, generator=g, dtype=torch.float64)), dones=dones,
                next_states=f32(torch.randn(n, S, generator=g, dtype=torch.float64)))


@pytest.mark.parametrize('S,lens,ep_len,truncated,normalize', [(2, [2000, 2000, 2000], 100, False, True), (2, [85, 60, 17, 1], 17, False, True),
                                                              (2, [120, 77], 25, True, False), (5, [300, 150], 50, True, True),
                                                              (8, [64], 16, False, True)])
def test_gae_kernel_matches_oracle(S, lens, ep_len, truncated, normalize):
    """mi_gae_advantages (returns -> LinearValue fit -> bootstraps -> GAE -> normalise, one workgroup per replay, fp64) against the
    oracle's restatement of compute_advantages / ch.normalize (rl.py:95-110,355) on each replay: ragged lengths, an early
    termination, a replay whose last episode is truncated, 2..8 state dimensions; and the fitted baseline weights."""
    from exploring_meta_amd.engine import gae_advantages
    from exploring_meta_amd.core_functions.rl import _device_batch
    gamma, tau, reg = 0.99, 0.95, 2.0
    eps = [_random_replay(40 + i, n, S, ep_len, truncated) for i, n in enumerate(lens)]
    batch = _device_batch(eps, S, 2, torch.device('cuda'))
    adv, wts = gae_advantages(batch['states'], batch['next_states'], batch['rewards'], batch['dones'], batch['count'], gamma, tau, reg,
                              normalize=normalize, want_weights=True)
    torch.cuda.synchronize()
    worst = 0.0
    for i, (ep, n) in enumerate(zip(eps, lens)):
        base = RL.LinearValue(S, reg)
        ref = RL.compute_advantages(base, tau, gamma, ep, True)
        if normalize:
            ref = RL.normalize(ref)
        ref = ref.reshape(-1).numpy()
        got = adv[i, :n].double().cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        worst = max(worst, float(np.abs(got - ref).max()) / scale)
        assert float(adv[i, n:].abs().sum()) == 0.0
        # the baseline's predictions (the weights themselves are ill-determined along near-null directions of F^T F)
        f = base._features(ep['states'])
        pred_ref, pred = (f @ base.weight).reshape(-1).numpy(), (f @ wts[i].cpu().reshape(-1, 1)).reshape(-1).numpy()
        assert np.abs(pred - pred_ref).max() <= 1e-6 * max(1.0, np.abs(pred_ref).max())
    report(f'gae_kernel[S{S},{lens}]', max_rel=worst)
    assert worst < 5e-7                               # fp32 output rounding


@pytest.mark.parametrize('lens', [[85, 60, 17, 1], [64, 64, 64], [2000] * 27 + [1731] * 110])
def test_packed_device_batch_matches_the_general_path(lens, monkeypatch):
    """_device_batch of replays whose fields already lie on the device as fp32 tensors (what the runners produce): mi_copy_segments packs
    all fields of all replays in one launch per 128 arrays, the row counts travel in kernel arguments (mi_upload_i32) -- against the
    general path (conversion chain per field, concatenate / pad / gather): identical tensors, zero padding rows; ragged and equal
    lengths, more than 128 arrays (685: six launches), and the same for the stacked parameters of the old policies."""
    from exploring_meta_amd.core_functions import rl as prl
    dev = torch.device('cuda', torch.cuda.current_device())
    S, A = 3, 2
    eps = [{k: v.float().to(dev) for k, v in _random_replay(7 + i, n, S, 16, False).items()} for i, n in enumerate(lens)]
    for e in eps:
        e['actions'] = torch.randn(e['states'].shape[0], A, device=dev)
    packed = prl._device_batch(eps, S, A, dev)
    assert prl._device_batch_packed(eps, lens, max(lens), S, A, dev) is not None          # the fast path is the one that ran
    monkeypatch.setattr(prl, '_device_batch_packed', lambda *a: None)
    plain = prl._device_batch(eps, S, A, dev)
    torch.cuda.synchronize()
    for k in ('states', 'actions', 'next_states', 'rewards', 'dones', 'count'):
        assert packed[k].shape == plain[k].shape and packed[k].dtype == plain[k].dtype and torch.equal(packed[k], plain[k]), k
    for i, n in enumerate(lens):
        assert float(packed['states'][i, n:].abs().sum()) == 0.0 and float(packed['rewards'][i, n:].abs().sum()) == 0.0
    # a Replay (what the runners return) remembers its checked addresses: a second gather reads them back, a changed entry forgets them
    monkeypatch.undo()
    reps = [prl.Replay(e) for e in eps]
    first = prl._device_batch(reps, S, A, dev)
    assert all(r._mi_pack is not None for r in reps)
    again = prl._device_batch(reps, S, A, dev)
    for k in ('states', 'actions', 'next_states', 'rewards', 'dones', 'count'):
        assert torch.equal(first[k], packed[k]) and torch.equal(again[k], packed[k]), k
    reps[0]['rewards'] = reps[0]['rewards'] * 2.0
    assert reps[0]._mi_pack is None
    changed = prl._device_batch(reps, S, A, dev)
    assert torch.equal(changed['rewards'][0, :lens[0]], 2.0 * packed['rewards'][0, :lens[0]]) and torch.equal(changed['rewards'][1:], packed['rewards'][1:])
    # a field that is not fp32 / contiguous / on the device sends the whole call down the general path
    eps[1]['rewards'] = eps[1]['rewards'].double()
    monkeypatch.undo()
    assert prl._device_batch_packed(eps, lens, max(lens), S, A, dev) is None
    # the stored policies' parameters, [tasks, P] in the engine's order
    pols = [cf.DiagNormalPolicy(2, 2).to(dev) for _ in range(min(len(lens), 30))]
    for i, q in enumerate(pols):
        with torch.no_grad():
            q.sigma.fill_(0.1 * i)
    got = prl._stacked_flat_parameters(pols, dev)
    assert got is not None and torch.equal(got, torch.stack([q.flat() for q in pols]))
    th = torch.randn(got.shape[1], device=dev)               # and back: load_flat scatters a vector into the parameters in one launch
    pols[1].load_flat(th)
    assert torch.equal(pols[1].flat(), th) and torch.equal(pols[1].sigma.detach(), th[:2])
    with pytest.raises(ValueError):
        pols[1].load_flat(th[:-1].contiguous())
    pols[0].double()
    assert prl._stacked_flat_parameters(pols, dev) is None


def test_fused_conjugate_gradient_update_on_long_vectors():
    """mi_cg_update against the recurrences in torch fp64 on vectors of the 2 x 100 policy's length (10,604) and longer ones (more
    elements than threads, lengths that are not multiples of the workgroup)."""
    from exploring_meta_amd import _lib
    from exploring_meta_amd.engine import _ptr, _stream
    lib = _lib.load()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    for n in (10604, 12288, 12289, 40000):
        r0 = torch.randn(n, generator=g, dtype=torch.float64)
        ap = (2.0 * r0 + 0.3 * torch.roll(r0, 1)).float()
        x, r, p = torch.zeros(n, dtype=torch.float64, device=dev), r0.to(dev), r0.to(dev)
        rr = torch.zeros(3, dtype=torch.float64, device=dev)
        rr[0] = torch.dot(r, r)
        p32 = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(lib.mi_cg_update(_stream(dev), _ptr(x), _ptr(r), _ptr(p), _ptr(ap.to(dev)), _ptr(rr), _ptr(p32), n, 1e-8))
        torch.cuda.synchronize()
        rr_old = torch.dot(r0, r0)
        alpha = rr_old / (torch.dot(r0, ap.double()) + 1e-8)
        x_ref, r_ref = alpha * r0, r0 - alpha * ap.double()
        rr_new = torch.dot(r_ref, r_ref)
        p_ref = r_ref + (rr_new / rr_old) * r0
        assert torch.allclose(x.cpu(), x_ref, rtol=1e-12, atol=1e-14) and torch.allclose(r.cpu(), r_ref, rtol=1e-12, atol=1e-14), n
        assert torch.allclose(p.cpu(), p_ref, rtol=1e-12, atol=1e-13) and torch.equal(p32.cpu(), p.cpu().float()), n
        assert abs(float(rr[0]) - float(rr_new)) <= 1e-12 * float(rr_new) and abs(float(rr[1]) - float(alpha)) <= 1e-12 * abs(float(alpha)), n


def test_solve_start_and_step_scaling_kernels():
    """mi_cg_init (x = 0, r = p = b, rr = (b.b, 0, 0)) and mi_trpo_scale_step (shs = 0.5 s.Fs, lagrange = sqrt(shs / max_kl), s / lagrange;
    reference rl.py:419-421) against the tensor expressions they replace."""
    from exploring_meta_amd import _lib
    from exploring_meta_amd.engine import _ptr, _stream
    lib = _lib.load()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(5)
    for n in (10604, 3, 40001):
        b = torch.randn(n, generator=g).to(dev)
        x, r, p = (torch.full((n,), 7.0, dtype=torch.float64, device=dev) for _ in range(3))
        p32, rr = torch.full((n,), 7.0, device=dev), torch.full((3,), 7.0, dtype=torch.float64, device=dev)
        _lib.check(lib.mi_cg_init(_stream(dev), _ptr(b), _ptr(x), _ptr(r), _ptr(p), _ptr(p32), _ptr(rr), n))
        assert float(x.abs().sum()) == 0.0 and torch.equal(r, b.double()) and torch.equal(p, b.double()) and torch.equal(p32, b)
        assert abs(float(rr[0]) - float(torch.dot(b.double(), b.double()))) <= 1e-12 * float(rr[0]) and float(rr[1]) == 0.0 and float(rr[2]) == 0.0
        fs = (2.0 * b + 0.1 * torch.roll(b, 1)).contiguous()
        out, lm = torch.empty_like(b), torch.empty(1, device=dev)
        _lib.check(lib.mi_trpo_scale_step(_stream(dev), _ptr(b), _ptr(fs), n, 0.01, _ptr(out), _ptr(lm)))
        lm_ref = torch.sqrt(0.5 * torch.dot(b.double(), fs.double()) / 0.01)
        assert abs(float(lm) - float(lm_ref)) <= 1e-6 * float(lm_ref)
        assert torch.allclose(out, (b.double() / lm_ref).float(), rtol=1e-6, atol=0)
    # a direction of negative curvature: NaN, as torch.sqrt of a negative number in the reference
    _lib.check(lib.mi_trpo_scale_step(_stream(dev), _ptr(b), _ptr((-fs).contiguous()), n, 0.01, _ptr(out), _ptr(lm)))
    assert torch.isnan(lm).all() and torch.isnan(out).all()


def test_surrogate_context_on_device_matches_host_path(monkeypatch):
    """_SurrogateContext built on the device (one mi_gae_advantages launch, no host round trip) == built by the host numpy walk."""
    from exploring_meta_amd.core_functions import rl as prl
    theta, replays, olds = _replays()
    pol = _policy(theta)
    old_pols = [_policy(o) for o in olds]
    dev_ctx = prl._SurrogateContext(replays, old_pols, pol, cf.LinearValue(2, 2), PARAMS)
    monkeypatch.setattr(prl, '_gae_on_device', lambda *a: False)
    host_base = cf.LinearValue(2, 2)
    host_ctx = prl._SurrogateContext(replays, old_pols, pol, host_base, PARAMS)
    for k in ('states', 'actions', 'adv', 'count'):
        a, b = dev_ctx.qry[k], host_ctx.qry[k]
        assert a.shape == b.shape and torch.allclose(a.float(), b.float(), rtol=0, atol=2e-6), k
        a, b = dev_ctx.sup[k], host_ctx.sup[k]
        assert a.shape == b.shape and torch.allclose(a.float(), b.float(), rtol=0, atol=2e-6), k
    l1, k1, g1 = dev_ctx.evaluate(pol.flat(), want_grad=True)
    l2, k2, g2 = host_ctx.evaluate(pol.flat(), want_grad=True)
    assert rel_err(g1.cpu().numpy(), g2.cpu().numpy()) < 1e-5 and abs(float(l1[0]) - float(l2[0])) < 1e-6


def test_per_task_losses_on_device_match_host_walk(monkeypatch):
    """trpo_update / trpo_a2c_loss (update_vf True and False: the query loss uses the baseline as fitted to the support replay,
    rl.py:401) with the advantages from mi_gae_advantages == with the host numpy walk."""
    from exploring_meta_amd.core_functions import rl as prl
    theta, replays, _ = _replays()
    pol = _policy(theta)
    cf.set_device(torch.device('cuda'))
    out = {}
    for mode in ('device', 'host'):
        if mode == 'host':
            monkeypatch.setattr(prl, '_gae_on_device', lambda *a: False)
        base = cf.LinearValue(2, 2)
        sup, qry = replays[1][0], replays[1][1]
        new = prl.trpo_update(sup, pol, base, 0.1, 0.99, 1.0)
        l_fit = prl.trpo_a2c_loss(sup, pol, base, 0.99, 1.0, update_vf=True)
        l_keep = prl.trpo_a2c_loss(qry, new, base, 0.99, 1.0, update_vf=False)
        out[mode] = (new.flat().detach().cpu().numpy(), float(l_fit), float(l_keep), np.array(base.weight).reshape(-1))
    assert rel_err(out['device'][0], out['host'][0]) < 1e-6
    assert abs(out['device'][1] - out['host'][1]) < 1e-6 and abs(out['device'][2] - out['host'][2]) < 1e-6
    f = cf.LinearValue(2, 2)._features(np.asarray(replays[1][0]['states'], dtype=np.float64))
    assert np.abs(f @ out['device'][3] - f @ out['host'][3]).max() < 1e-6


def test_fused_conjugate_gradient_matches_the_plain_loop():
    """mi_cg_update (the recurrences of cherry's conjugate_gradient as one launch per iteration) against the same loop written with
    torch operations, on a random symmetric positive definite system: same iterates, same result at the early exit."""
    from exploring_meta_amd.core_functions import rl as prl
    g = torch.Generator().manual_seed(9)
    n = 700
    M = torch.randn(n, n, generator=g, dtype=torch.float64) / n ** 0.5
    A = (M.t() @ M + 0.5 * torch.eye(n, dtype=torch.float64))
    b = torch.randn(n, generator=g, dtype=torch.float64).float()
    A32 = A.float()
    ref = prl.conjugate_gradient(lambda v: A32 @ v, b, num_iterations=10)                       # CPU tensors: the plain loop
    A_dev, calls = A32.cuda(), []

    def Ax(v):
        calls.append(1)
        return A_dev @ v

    got = prl.conjugate_gradient(Ax, b.cuda(), num_iterations=10)
    assert len(calls) == 10 and rel_err(got.cpu().numpy(), ref.numpy()) < 1e-5                 # (fp32 products on two devices)
    # early exit: a well conditioned system converges below tol before the iteration cap.  The plain loop breaks; the device loop takes
    # the break on the device (mi_cg_update_checked latches "converged", later recurrences are no-ops: no host synchronisation per
    # iteration) -- the iterate returned is the iterate AT the break, whatever the products computed afterwards
    A2 = (torch.eye(n, dtype=torch.float64) * 2.0).float()
    ref2 = prl.conjugate_gradient(lambda v: A2 @ v, b, num_iterations=10)
    calls.clear()
    A2d = A2.cuda()
    got2 = prl.conjugate_gradient(lambda v: (calls.append(1), A2d @ v * (1.0 if len(calls) == 1 else 7.0))[1], b.cuda(), num_iterations=10)
    assert torch.allclose(got2.cpu(), b / 2.0, rtol=1e-6, atol=1e-7) and torch.allclose(got2.cpu(), ref2, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('case', ['small_T4_B150', 'ragged_counts', 'cfg5_T20_B2000'])
def test_fused_fvp_sweeps_match_the_per_layer_path_and_the_oracle(case):
    """mi_trpo_fvp as three fused sweeps + folds (csrc/policy_sweep.h, the default for the reference's 2-100-100-2 ReLU policy)
    against the per-layer launch sequence it replaces (mi_policy_set_fused_fvp(0)) on the same stored passes: a batch that is not
    a multiple of the 32-row slab, tasks with fewer valid rows than the padded batch (count < B: padding rows must contribute
    nothing), and BASELINE config 5's size (20 tasks x 2000 rows: five slabs per workgroup, workgroups that straddle two tasks).
    Same mathematics in another summation order: agreement to fp32 rounding; and (small case) against the fp64 oracle's
    hessian_vector_product of the mean KL."""
    from exploring_meta_amd import _lib
    from exploring_meta_amd.core_functions.rl import _SurrogateContext
    lib = _lib.load()
    params = PARAMS_CFG5 if case.startswith('cfg5') else PARAMS
    theta, replays, olds = _replays(params)
    pol = _policy(theta)
    ctx = _SurrogateContext(replays, [_policy(o) for o in olds], pol, cf.LinearValue(2, 2), params)
    if case == 'ragged_counts':          # shorten some tasks' valid rows: the rows behind count are padding to every kernel
        for d in (ctx.sup, ctx.qry):
            c = d['count'].clone()
            c[0], c[2] = 97, 33
            d['count'] = c.contiguous()
    th = pol.flat()
    g = torch.Generator(device='cuda').manual_seed(11)
    outs, evals = {}, {}
    cand = th + 0.01 * torch.sin(torch.arange(th.numel(), device='cuda', dtype=torch.float32))      # a displaced line-search candidate
    try:
        for fused in (0, 1):
            lib.mi_policy_set_fused_fvp(fused)
            # the surrogate itself: inner step + query pass as two primal sweeps (fused) vs the per-layer launches; then a forward-only
            # evaluation at a displaced candidate (the line search's call), then back to theta so that the products below see its passes
            l1, k1, g1 = ctx.evaluate(th, want_grad=True)
            l2, k2, _ = ctx.evaluate(cand)
            evals[fused] = (float(l1), float(k1), g1.clone(), float(l2), float(k2))
            ctx.evaluate(th, want_grad=True)
            res = []
            for k in range(3):
                v = torch.randn(th.numel(), device='cuda', generator=torch.Generator(device='cuda').manual_seed(20 + k))
                res.append(ctx.fvp(th, v).clone())
            torch.cuda.synchronize()
            outs[fused] = res
    finally:
        lib.mi_policy_set_fused_fvp(1)
    errs = [rel_err(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(outs[1], outs[0])]
    rep = dict(fused_vs_per_layer_rel=errs)
    (l1a, k1a, g1a, l2a, k2a), (l1b, k1b, g1b, l2b, k2b) = evals[0], evals[1]
    rep['surrogate_grad_fused_vs_per_layer_rel'] = rel_err(g1b.cpu().numpy(), g1a.cpu().numpy())
    rep['surrogate_loss_abs_diff'] = [abs(l1a - l1b), abs(l2a - l2b)]
    rep['kl_abs_diff'] = [abs(k1a - k1b), abs(k2a - k2b)]
    assert rep['surrogate_grad_fused_vs_per_layer_rel'] < 2e-5
    assert abs(l1a - l1b) < 1e-6 * max(1.0, abs(l1a)) and abs(l2a - l2b) < 2e-6 * max(1.0, abs(l2a))
    assert abs(k1a - k1b) < 1e-7 and abs(k2a - k2b) < 1e-6 * max(k2a, 1e-3)
    assert all(torch.isfinite(x).all() for x in outs[1]) and max(errs) < 2e-5, errs
    # deterministic: a repeated fused product is bit-identical (fixed-order folds, no atomics)
    v = torch.randn(th.numel(), device='cuda', generator=torch.Generator(device='cuda').manual_seed(20))
    assert torch.equal(ctx.fvp(th, v), outs[1][0])
    if case == 'small_T4_B150':
        p64 = OrderedDict((k, x.clone().requires_grad_(True)) for k, x in theta.items())
        _, kl = RL.meta_surrogate_loss(replays, olds, p64, RL.LinearValue(2, 2), params)
        Fvp = RL.hessian_vector_product(kl, list(p64.values()))
        v64 = torch.randn(th.numel(), device='cuda', generator=torch.Generator(device='cuda').manual_seed(20)).double().cpu()
        rep['fused_vs_oracle_rel'] = rel_err(outs[1][0].cpu().numpy(), Fvp(v64).detach().numpy())
        assert rep['fused_vs_oracle_rel'] < 1e-3
    report(f'fused_fvp[{case}]', **rep)


@pytest.mark.parametrize('S,A,T,B', [(3, 3, 3, 70), (4, 6, 2, 33), (1, 1, 5, 64), (2, 4, 7, 129)])
def test_fused_policy_sweeps_other_state_and_action_widths(S, A, T, B):
    """The fused sweeps keep their per-row tables at compile-time pitches (4 states, 6 actions) and walk the action dimensions in
    pairs: policies with other state / action widths (odd action counts, the maxima, width 1) on synthetic replays, surrogate loss /
    KL / gradient and three Fisher-vector products against the per-layer path (mi_policy_set_fused_fvp(0)) on the same inputs."""
    from exploring_meta_amd import _lib
    from exploring_meta_amd.engine import PolicyEngine
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(100 * S + A)
    eng = PolicyEngine(S, A, (100, 100), activation='relu')
    P = eng.param_count
    rnd = lambda *shape: torch.randn(*shape, device='cuda', generator=g)
    theta = 0.1 * rnd(P)
    theta[:A] = torch.linspace(-0.4, 0.3, A, device='cuda')                      # sigma
    count = torch.randint(B // 2, B + 1, (T,), device='cuda', generator=g, dtype=torch.int32)
    count[0] = B

    def replay():
        d = dict(states=rnd(T, B, S).contiguous(), actions=rnd(T, B, A).contiguous(), adv=rnd(T, B).contiguous(), count=count)
        rows = torch.arange(B, device='cuda')[None, :] >= count[:, None]                  # padding rows are zeros, as the drivers pad
        for k in ('states', 'actions'):
            d[k][rows] = 0.0
        d['adv'][rows] = 0.0
        return d

    sup, qry = replay(), replay()
    old_loc = (0.3 * rnd(T, B, A)).contiguous()
    old_scale = (0.5 + torch.rand(T, A, device='cuda', generator=g)).contiguous()
    v = [rnd(P) for _ in range(3)]
    out = {}
    try:
        for fused in (0, 1):
            lib.mi_policy_set_fused_fvp(fused)
            loss, kl, grad = eng.surrogate(theta, sup, qry, old_loc, old_scale, 0.1, True)
            fv = [eng.fvp(theta, sup, qry, 0.1, 1e-5, x).clone() for x in v]
            l2, k2, _ = eng.surrogate(theta + 0.01, sup, qry, old_loc, old_scale, 0.1, False)
            torch.cuda.synchronize()
            out[fused] = (float(loss), float(kl), grad.clone(), fv, float(l2), float(k2))
    finally:
        lib.mi_policy_set_fused_fvp(1)
    a, b = out[0], out[1]
    eg = rel_err(b[2].cpu().numpy(), a[2].cpu().numpy())
    ef = [rel_err(y.cpu().numpy(), x.cpu().numpy()) for x, y in zip(a[3], b[3])]
    report(f'fused_sweeps_widths[S{S},A{A},T{T},B{B}]', loss=[a[0], b[0]], kl=[a[1], b[1]], grad_rel=eg, fvp_rel=ef)
    assert all(torch.isfinite(x).all() for x in [b[2]] + b[3])
    assert abs(a[0] - b[0]) < 2e-6 * max(1.0, abs(a[0])) and abs(a[1] - b[1]) < 2e-6 * max(1.0, abs(a[1]))
    assert abs(a[4] - b[4]) < 2e-6 * max(1.0, abs(a[4])) and abs(a[5] - b[5]) < 2e-6 * max(1.0, abs(a[5]))
    assert eg < 2e-5 and max(ef) < 2e-5, (eg, ef)
