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
