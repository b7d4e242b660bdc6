"""oracle/rl_ref.py: the policy against fixtures from the reference's own DiagNormalPolicy (G5), and self-consistency of the
restated TRPO machinery (the cherry / learn2learn pieces have no reference-side fixtures: parity unpinned there)."""
from collections import OrderedDict

import numpy as np
import torch

from oracle import rl_ref as RL
from helpers import hash_params


def _policy(dtype=torch.float64):
    p = hash_params(RL.policy_param_shapes(), 19, dtype)
    p['sigma'] = torch.tensor([-0.3, 0.2], dtype=dtype)
    return p


def test_param_order_matches_reference(golden_small):
    assert list(RL.policy_param_shapes().keys()) == list(golden_small['g5_param_names'])


def test_policy_log_prob_and_grad_match_reference(golden_small):
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in _policy().items())
    st, ac = torch.from_numpy(golden_small['g5_states']), torch.from_numpy(golden_small['g5_actions'])
    loc, scale = RL.policy_loc_scale(p, st)
    lp = RL.policy_log_prob(p, st, ac)
    assert np.allclose(loc.detach().numpy(), golden_small['g5_policy_f64_loc'], rtol=1e-12, atol=1e-14)
    assert np.allclose(scale.detach().numpy(), golden_small['g5_policy_f64_scale'], rtol=1e-12)
    assert np.allclose(lp.detach().numpy(), golden_small['g5_policy_f64_logp'], rtol=1e-11, atol=1e-13)
    g = torch.autograd.grad(lp.sum(), list(p.values()))
    assert np.allclose(torch.cat([x.reshape(-1) for x in g]).numpy(), golden_small['g5_policy_f64_grad'], rtol=1e-9, atol=1e-12)


def test_anil_policy_matches_reference(golden_small):
    """DiagNormalPolicyANIL (tanh body) from the reference: log-prob, and the gradient with the body grads on / off
    (policies.py:97-106) -- the `head_only` semantics of the oracle's trpo_update."""
    raw = hash_params(RL.anil_policy_param_shapes(), 23)
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in RL.anil_as_policy_params(raw).items())
    assert list(p.keys()) == list(RL.policy_param_shapes().keys())
    st, ac = torch.from_numpy(golden_small['g5_states']), torch.from_numpy(golden_small['g5_actions'])
    lp = RL.policy_log_prob(p, st, ac, activation=torch.tanh)
    for off in (0, 1):
        assert np.allclose(lp.detach().numpy(), golden_small[f'g5_anil_f64_bodyoff{off}_logp'], rtol=1e-11, atol=1e-13)
    g = torch.autograd.grad(lp.sum(), list(p.values()))
    flat = torch.cat([x.reshape(-1) for x in g]).numpy()
    assert np.allclose(flat, golden_small['g5_anil_f64_bodyoff0_grad'], rtol=1e-9, atol=1e-12)
    head = np.concatenate([np.ones(v.numel()) if (k == 'sigma' or k.startswith('mean.4.')) else np.zeros(v.numel()) for k, v in p.items()])
    assert np.allclose(flat * head, golden_small['g5_anil_f64_bodyoff1_grad'], rtol=1e-9, atol=1e-12)
    # head_only update = that masked gradient
    ep = dict(states=st, actions=ac, rewards=torch.from_numpy(golden_small['g5_actions'][:, :1]), dones=torch.zeros(64, 1, dtype=torch.float64),
              next_states=st)
    ep['dones'][-1] = 1.0
    new = RL.trpo_update(ep, p, RL.LinearValue(2, 2), 0.1, 0.99, 1.0, first_order=True, activation=torch.tanh, head_only=True)
    for k in p:
        moved = not torch.equal(new[k], p[k])
        assert moved == (k == 'sigma' or k.startswith('mean.4.')), k


def test_discount_and_gae():
    r = torch.tensor([[1.0], [1.0], [1.0], [2.0]], dtype=torch.float64)
    d = torch.tensor([[0.0], [1.0], [0.0], [1.0]], dtype=torch.float64)
    assert torch.allclose(RL.discount(0.5, r, d), torch.tensor([[1.5], [1.0], [2.0], [2.0]], dtype=torch.float64))
    v = torch.zeros(4, 1, dtype=torch.float64)
    adv = RL.generalized_advantage(0.5, 1.0, r, d, v, torch.zeros(1))
    assert torch.allclose(adv, RL.discount(0.5, r, d))            # V == 0 and tau == 1: GAE reduces to the return


def test_cg_solves_spd_system():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(12, 12, generator=g, dtype=torch.float64)
    a = a @ a.t() + 0.5 * torch.eye(12, dtype=torch.float64)
    b = torch.randn(12, generator=g, dtype=torch.float64)
    x = RL.conjugate_gradient(lambda v: a @ v, b, num_iterations=50)
    assert torch.allclose(a @ x, b, atol=1e-6)


def test_meta_iteration_runs_and_improves_surrogate():
    torch.manual_seed(0)
    env = RL.Particles2D(seed=1)
    gen = torch.Generator().manual_seed(2)
    params = dict(inner_lr=0.1, max_path_length=20, adapt_steps=1, adapt_batch_size=4, meta_batch_size=3, outer_lr=0.3,
                  backtrack_factor=0.5, ls_max_steps=15, max_kl=0.01, tau=1.0, gamma=0.99)
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in _policy().items())
    baseline = RL.LinearValue(2, 2)
    replays, pols = [], []
    for task in env.sample_tasks(params['meta_batch_size']):
        env.set_task(task)
        learner = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in p.items())
        adapted, vloss, rep, rew = RL.fast_adapt_trpo(env, learner, baseline, params, gen, first_order=True)
        replays.append(rep)
        pols.append(OrderedDict((k, v.detach()) for k, v in adapted.items()))
    out = RL.meta_optimize_trpo(params, p, baseline, replays, pols)
    assert out['accepted'] is not None and out['new_loss'] < out['old_loss'] and out['kl'] < params['max_kl']
    # at the old parameters the adapted policy equals the stored one: KL == 0 and its gradient vanishes, so the Fisher
    # product is symmetric positive semi-definite
    v = torch.randn(out['grad'].shape, dtype=torch.float64)
    assert torch.dot(v, out['fvp'](v)) > 0


def test_dice_restatement_equals_the_literal_in_place_loop():
    """oracle dice_log_probs (out of place, differentiable) against the reference's literal lines rl.py:202-205,219-225 executed
    in place -- including the wrap-around of values[i - 1] at i = 0 -- and magic_box's value / derivative."""
    torch.manual_seed(0)
    n = 23
    lp = torch.randn(n, 1, dtype=torch.float64)
    dones = torch.zeros(n, 1, dtype=torch.float64)
    dones[[6, 14, 22]] = 1

    weights = torch.ones_like(dones)                       # rl.py:220-222
    weights[1:].add_(dones[:-1], alpha=-1.0)
    weights /= dones.sum()
    values = lp.clone()
    for i in range(values.size(0)):                        # rl.py:203-204
        values[i] += values[i - 1] * weights[i]

    x = lp.clone().requires_grad_(True)
    out = RL.dice_log_probs(x, dones)
    assert torch.equal(out.detach(), torch.ones_like(out))                         # magic_box evaluates to 1
    k = torch.randn(n, 1, dtype=torch.float64)
    g = torch.autograd.grad((out * k).sum(), x)[0]                                   # = M^T k
    # M from the literal loop: column j of M = the loop applied to the j-th unit vector (the recurrence is linear)
    M = torch.zeros(n, n, dtype=torch.float64)
    for j in range(n):
        e = torch.zeros(n, 1, dtype=torch.float64)
        e[j] = 1
        for i in range(n):
            e[i] += e[i - 1] * weights[i]
        M[:, j] = e[:, 0]
    assert torch.allclose(M @ lp, values)
    assert torch.allclose(g, M.t() @ k)
    assert M[0, n - 1] != 0                                                          # the wrap-around term is there


# ------------------------------------------------------------------------------------ composition pinned to the reference's own lines
import pytest                                                                    # noqa: E402
import rl_cases                                                                  # noqa: E402


def _flat(p):
    return torch.cat([v.detach().reshape(-1) for v in (p.values() if hasattr(p, 'values') else p)]).numpy()


def _close(a, b, tol=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b)) <= tol * max(1.0, float(np.max(np.abs(b)))), float(np.max(np.abs(a - b)))


@pytest.mark.parametrize('name', list(rl_cases.CASES))
def test_composition_matches_the_reference(golden_rl, name, monkeypatch):
    """oracle/rl_ref.py against tests/golden/golden_rl.npz = the REFERENCE's compute_advantages (rl.py:95-110), trpo_a2c_loss
    (:346-358), trpo_update (:361-374), fast_adapt_trpo (:377-406), meta_surrogate_loss (:441-473), meta_optimize_trpo (:409-438)
    executed in the build container with this oracle's leaf restatements standing in for cherry / learn2learn: every detach,
    refit, mean, KL direction, step scaling and line-search test of those lines, to 1e-9 in fp64."""
    g = golden_rl
    case = rl_cases.load_case(g, name)
    params, theta, replays, olds, act, anil = (case[k] for k in ('params', 'theta', 'replays', 'olds', 'activation', 'anil'))
    pre = f'rl_{name}_f64'
    leaf = lambda: OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())

    # rl.py:95-110
    bl = RL.LinearValue(2, 2)
    _close(RL.compute_advantages(bl, params['tau'], params['gamma'], replays[0][0]).numpy(), g[f'{pre}_adv_fit'])
    _close(RL.compute_advantages(bl, params['tau'], params['gamma'], replays[0][-1], update_vf=False).numpy(), g[f'{pre}_adv_nofit'])
    _close(bl.weight.numpy(), g[f'{pre}_vf_weight'], 1e-7)          # (normal equations: the solution itself is conditioned ~1e7)

    # rl.py:346-374
    p = leaf()
    loss = RL.trpo_a2c_loss(replays[0][0], p, RL.LinearValue(2, 2), params['gamma'], params['tau'], activation=act)
    _close([loss.item()], g[f'{pre}_inner_loss'])
    _close(_flat(torch.autograd.grad(loss, list(p.values()))), g[f'{pre}_inner_grad'])
    new = RL.trpo_update(replays[0][0], leaf(), RL.LinearValue(2, 2), params['inner_lr'], params['gamma'], params['tau'], first_order=True,
                         activation=act)
    _close(_flat(new), g[f'{pre}_adapted_theta'])

    # rl.py:377-406 (head-only inner steps under anil; validation loss without a refit; success rate from the success flags)
    n_q = replays[0][-1]['states'].shape[0]
    has = bool(g[f'{pre}_fa_has_success'][0])
    succ = rl_cases.success_flags(n_q) if has else None
    if has:
        assert np.array_equal(succ.numpy(), g[f'{pre}_fa_success_flags'])
    adapted, vloss, rew, suc = RL.fast_adapt_trpo_replayed(replays[0], leaf(), RL.LinearValue(2, 2), params, first_order=True, activation=act,
                                                           anil=anil, success=succ)
    _close(_flat(adapted), g[f'{pre}_fa_theta'])
    _close([vloss.item()], g[f'{pre}_fa_valid_loss'])
    _close([rew, suc], g[f'{pre}_fa_reward_success'])
    if anil:                                                         # the body did not move (rl.py:381-382 + allow_unused)
        body = slice(2, 2 + 100 * 2 + 100 + 100 * 100 + 100)
        assert np.array_equal(_flat(adapted)[body], _flat(theta)[body])

    # rl.py:441-473 and the gradient of :413-416
    p = leaf()
    sl, kl = RL.meta_surrogate_loss(replays, olds, p, RL.LinearValue(2, 2), params, act)
    _close([sl.item(), kl.item()], g[f'{pre}_surr_loss_kl'])
    _close(_flat(torch.autograd.grad(sl, list(p.values()), retain_graph=True)), g[f'{pre}_surr_grad'])
    cand = OrderedDict((k, (v.detach() + 0.01 * torch.sin(torch.arange(v.numel(), dtype=torch.float64)).view_as(v)).requires_grad_(True))
                       for k, v in theta.items())
    sl2, kl2 = RL.meta_surrogate_loss(replays, olds, cand, RL.LinearValue(2, 2), params, act)
    _close([sl2.item(), kl2.item()], g[f'{pre}_surr_displaced_loss_kl'])

    # rl.py:409-438: every Fisher-vector product (10 CG iterations + shs) and every evaluation of the line search
    fin, fout, evals = [], [], []
    hvp0, msl0 = RL.hessian_vector_product, RL.meta_surrogate_loss

    def hvp_rec(loss_, ps_, **k):
        f = hvp0(loss_, ps_, **k)
        def call(v):
            r = f(v)
            fin.append(v.detach().clone())
            fout.append(r.detach().clone())
            return r
        return call

    def msl_rec(*a, **k):
        l_, k_ = msl0(*a, **k)
        evals.append((l_.item(), k_.item()))
        return l_, k_
    monkeypatch.setattr(RL, 'hessian_vector_product', hvp_rec)
    monkeypatch.setattr(RL, 'meta_surrogate_loss', msl_rec)
    p = leaf()
    out = RL.meta_optimize_trpo(params, p, RL.LinearValue(2, 2), replays, olds, act)
    tol = 1e-7 if anil else 1e-9        # (ANIL: ten CG iterations on the indefinite exact KL Hessian amplify the last bits ~1e3-fold)
    assert len(fout) == int(g[f'{pre}_opt_n_fvp'][0]) == 11
    assert (-1 if out['accepted'] is None else out['accepted']) == int(g[f'{pre}_opt_accepted'][0])
    _close(np.array(evals), g[f'{pre}_opt_evals'], tol)
    _close(np.array([[torch.dot(a, b).item(), b.norm().item()] for a, b in zip(fin, fout)]), g[f'{pre}_opt_fvp_dots'], tol)
    _close(fout[0].numpy(), g[f'{pre}_opt_fvp_first'])
    _close(fout[-1].numpy(), g[f'{pre}_opt_fvp_last'], tol)
    _close(fin[-1].numpy(), g[f'{pre}_opt_cg_step'], tol)
    _close(_flat(p), g[f'{pre}_opt_theta_new'], tol)


def test_reference_fp32_run_is_recorded_next_to_its_fp64_run(golden_rl):
    """What "fp32 parity" can mean on this path: the reference's OWN fp32 run against its fp64 run (advantages, inner gradient,
    surrogate gradient, CG step, new parameters).  At config-5 size the fp32 normal equations of the baseline fit (features up to
    t^3 = 8000) lose the advantages altogether; the HIP path fits in fp64 on the device and is held to the fp64 record."""
    small = golden_rl['rl_small_relu_f32_rel_to_f64']
    assert small[:3].max() < 1e-4
    assert golden_rl['rl_cfg5_f32_rel_to_f64'][0] > 0.05
