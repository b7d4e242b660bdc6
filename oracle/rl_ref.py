"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the MAML-TRPO path (reference core_functions/rl.py TRPO part,
core_functions/policies.py, rl/maml_trpo.py) for BASELINE config 5 (Particles2D, 2x100 MLP policy).

Pinned: ``DiagNormalPolicy`` / ``DiagNormalPolicyANIL`` density and log_prob against fixtures produced by the reference's own
classes (tests/golden G5).  COMPOSITION PINNED (round 6): ``compute_advantages``, ``trpo_a2c_loss``, ``trpo_update``, ``fast_adapt_trpo``,
``meta_surrogate_loss`` and ``meta_optimize_trpo`` below are held to 1e-9 (fp64) against ``tests/golden/golden_rl.npz``, which
``tests/golden/make_golden_rl.py`` records by EXECUTING the reference's own rl.py:95-110,346-473 with this file's leaf restatements
installed in place of the absent cherry / learn2learn leaves (``tests/test_oracle_rl.py::test_composition_matches_the_reference``).
LEAVES PARITY UNPINNED (third-party code absent from /root/reference, versions unpinned in
requirements.txt:6): cherry-rl's ``td.discount``, ``pg.generalized_advantage``, ``normalize``, ``models.robotics.LinearValue``,
``a2c.policy_loss``, ``trpo.policy_loss``, ``trpo.hessian_vector_product``, ``trpo.conjugate_gradient``, and learn2learn's
``Particles2D`` / ``clone_module`` / ``maml_update`` -- restated below from their published behaviour, anchored on the
reference's call sites (cited per function).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

EPSILON = 1e-6          # policies.py:14


# ----------------------------------------------------------------------------------------------- policy (pinned by G5)
def policy_param_shapes(input_size=2, output_size=2, hiddens=(100, 100)):
    """DiagNormalPolicy registration order (policies.py:30-47): mean.{0,2,4}.{weight,bias} then sigma -- as named_parameters()
    yields them: 'sigma' first (registered after the Sequential? no: nn.Module lists direct Parameters before sub-modules)."""
    shapes = OrderedDict()
    shapes['sigma'] = (output_size,)
    sizes = [input_size] + list(hiddens) + [output_size]
    for i in range(len(sizes) - 1):
        shapes[f'mean.{2 * i}.weight'] = (sizes[i + 1], sizes[i])
        shapes[f'mean.{2 * i}.bias'] = (sizes[i + 1],)
    return shapes


def anil_policy_param_shapes(input_size=2, output_size=2, fc_neurons=100, hiddens=(100, 100)):
    """DiagNormalPolicyANIL (policies.py:70-95) named_parameters() order: the module's own `sigma` first, then body.*, head.*."""
    shapes = OrderedDict()
    shapes['sigma'] = (output_size,)
    sizes = [input_size] + list(hiddens)
    for i in range(len(sizes) - 1):
        shapes[f'body.{2 * i}.weight'] = (sizes[i + 1], sizes[i])
        shapes[f'body.{2 * i}.bias'] = (sizes[i + 1],)
    shapes['head.weight'] = (output_size, fc_neurons)
    shapes['head.bias'] = (output_size,)
    return shapes


def anil_as_policy_params(p):
    """body.{i}.* / head.* -> mean.{i}.* / mean.{last}.*: the ANIL policy is the same MLP with tanh between the layers
    (forward_pass = head(body(state)), policies.py:100-106); use with activation=torch.tanh."""
    n_body = sum(1 for k in p if k.startswith('body.') and k.endswith('.weight'))
    out = OrderedDict()
    for k, v in p.items():
        if k.startswith('body.'):
            out['mean.' + k[len('body.'):]] = v
        elif k.startswith('head.'):
            out[f'mean.{2 * n_body}.' + k[len('head.'):]] = v
        else:
            out[k] = v
    return out


def policy_loc_scale(p, state, activation=torch.relu):
    """density (policies.py:49-52): loc = MLP(state), scale = exp(clamp(sigma, min=log(EPSILON)))."""
    h = state
    n_lin = (len(p) - 1) // 2
    for i in range(n_lin):
        h = torch.nn.functional.linear(h, p[f'mean.{2 * i}.weight'], p[f'mean.{2 * i}.bias'])
        if i < n_lin - 1:
            h = activation(h)
    scale = torch.exp(torch.clamp(p['sigma'], min=math.log(EPSILON)))
    return h, scale


def normal_log_prob(loc, scale, value):
    var = scale ** 2
    return -((value - loc) ** 2) / (2 * var) - torch.log(scale) - math.log(math.sqrt(2 * math.pi))


def policy_log_prob(p, state, action, activation=torch.relu):
    """log_prob (policies.py:54-56): Normal.log_prob(action).mean(dim=1, keepdim=True)."""
    loc, scale = policy_loc_scale(p, state, activation)
    return normal_log_prob(loc, scale, action).mean(dim=1, keepdim=True)


def normal_kl(loc_p, scale_p, loc_q, scale_q):
    """torch.distributions.kl_divergence(Normal p, Normal q) elementwise."""
    var_ratio = (scale_p / scale_q) ** 2
    t1 = ((loc_p - loc_q) / scale_q) ** 2
    return 0.5 * (var_ratio + t1 - 1 - torch.log(var_ratio))


# ----------------------------------------------------------------------------------------------- cherry (UNPINNED)
def discount(gamma, rewards, dones):
    """cherry.td.discount: R_t = r_t + gamma (1 - d_t) R_{t+1} (call site rl.py:96)."""
    R = torch.zeros_like(rewards[0])
    out = torch.zeros_like(rewards)
    for t in reversed(range(rewards.shape[0])):
        R = rewards[t] + gamma * (1.0 - dones[t]) * R
        out[t] = R
    return out


def generalized_advantage(gamma, tau, rewards, dones, values, next_value):
    """cherry.pg.generalized_advantage (call site rl.py:105-110): delta_t = r_t + gamma (1-d_t) V_{t+1} - V_t,
    A = discount(gamma*tau, delta, dones)."""
    next_values = torch.cat([values[1:], next_value.reshape(1, 1).to(values.dtype)], dim=0)
    td = rewards + gamma * (1.0 - dones) * next_values - values
    return discount(gamma * tau, td, dones)


def normalize(x, epsilon=1e-8):
    """cherry.normalize (call site rl.py:355): (x - mean) / (std + eps), unbiased std."""
    if x.numel() <= 1:
        return x
    return (x - x.mean()) / (x.std() + epsilon)


class LinearValue:
    """cherry.models.robotics.LinearValue(input_size, reg) (call site maml_trpo.py:85 passes action_size as ``reg``):
    features [s, s^2, t, t^2, t^3, 1] with t = arange(T)/100 per episode batch; ridge least squares."""

    def __init__(self, input_size, reg=1e-5):
        self.input_size, self.reg = input_size, reg
        self.weight = torch.zeros(2 * input_size + 4, 1, dtype=torch.float64)

    def _features(self, states):
        length = states.shape[0]
        ones = torch.ones(length, 1, dtype=states.dtype)
        al = torch.arange(length, dtype=states.dtype).view(-1, 1) / 100.0
        return torch.cat([states, states ** 2, al, al ** 2, al ** 3, ones], dim=1)

    def fit(self, states, returns):
        f = self._features(states)
        reg = self.reg * torch.eye(f.shape[1], dtype=f.dtype)
        a = f.t() @ f + reg
        b = f.t() @ returns
        self.weight = torch.linalg.lstsq(a, b).solution

    def __call__(self, states):
        return self._features(states) @ self.weight.to(states.dtype)


def a2c_policy_loss(log_probs, advantages):
    """cherry.algorithms.a2c.policy_loss (rl.py:358): -mean(log_probs * advantages)."""
    return -torch.mean(log_probs * advantages)


def trpo_policy_loss(new_log_probs, old_log_probs, advantages):
    """cherry.algorithms.trpo.policy_loss (rl.py:469): -mean(exp(new - old) * advantages)."""
    return -torch.mean(torch.exp(new_log_probs - old_log_probs) * advantages)


def ppo_policy_loss(new_log_probs, old_log_probs, advantages, clip=0.1):
    """cherry.algorithms.ppo.policy_loss (rl.py:290,312,333): -mean(min(r A, clamp(r, 1-clip, 1+clip) A)), r = exp(new - old)."""
    ratios = torch.exp(new_log_probs - old_log_probs)
    obj = ratios * advantages
    obj_clip = ratios.clamp(1.0 - clip, 1.0 + clip) * advantages
    return -torch.min(obj, obj_clip).mean()


def conjugate_gradient(Ax, b, num_iterations=10, tol=1e-10, eps=1e-8):
    """cherry.algorithms.trpo.conjugate_gradient (rl.py:418)."""
    x = torch.zeros_like(b)
    r = b.clone()
    p = r.clone()
    r_dot_old = torch.dot(r, r)
    for _ in range(num_iterations):
        Ap = Ax(p)
        alpha = r_dot_old / (torch.dot(p, Ap) + eps)
        x = x + alpha * p
        r = r - alpha * Ap
        r_dot_new = torch.dot(r, r)
        p = r + (r_dot_new / r_dot_old) * p
        r_dot_old = r_dot_new
        if r_dot_new.item() < tol:
            break
    return x


def hessian_vector_product(loss, parameters, damping=1e-5):
    """cherry.algorithms.trpo.hessian_vector_product (rl.py:417)."""
    parameters = list(parameters)
    grad = torch.autograd.grad(loss, parameters, create_graph=True, retain_graph=True)
    flat = torch.cat([g.reshape(-1) for g in grad])

    def hvp(v):
        prod = torch.dot(flat, v)
        h = torch.autograd.grad(prod, parameters, retain_graph=True)
        return torch.cat([x.reshape(-1) for x in h]) + damping * v
    return hvp


# ----------------------------------------------------------------------------------------------- environment (UNPINNED)
class Particles2D:
    """learn2learn.gym.envs.Particles2D: state in R^2 starts at 0, goal ~ U(-0.5, 0.5)^2, action clipped to +-0.1,
    reward = -||state - goal||_2, done when both |state - goal| < 0.01."""

    def __init__(self, seed=0):
        self.rng = np.random.RandomState(seed)
        self.goal = np.zeros(2, dtype=np.float32)
        self.state = np.zeros(2, dtype=np.float32)
        self.state_size, self.action_size = 2, 2

    def sample_tasks(self, num_tasks):
        goals = self.rng.uniform(-0.5, 0.5, size=(num_tasks, 2))
        return [{'goal': g} for g in goals]

    def set_task(self, task):
        self.goal = np.asarray(task['goal'], dtype=np.float32)

    def reset(self):
        self.state = np.zeros(2, dtype=np.float32)
        return self.state.copy()

    def step(self, action):
        action = np.clip(action, -0.1, 0.1)
        self.state = self.state + action
        dx, dy = self.state - self.goal
        reward = -math.sqrt(dx * dx + dy * dy)
        done = abs(dx) < 0.01 and abs(dy) < 0.01
        return self.state.copy(), reward, done, self.goal


def collect_episodes(env, p, episodes, max_path_length, generator, dtype=torch.float64, activation=torch.relu):
    """Stand-in for core_functions/runner.py (cherry Runner fork, out of scope): ``episodes`` full episodes of at most
    ``max_path_length`` steps, actions sampled from the policy (policies.py:58-61).  Returns a dict of [N,*] tensors."""
    S, A, Rw, D, NS = [], [], [], [], []
    with torch.no_grad():
        for _ in range(episodes):
            s = env.reset()
            for t in range(max_path_length):
                st = torch.as_tensor(s, dtype=dtype).view(1, -1)
                loc, scale = policy_loc_scale(p, st, activation)
                a = (loc + scale * torch.randn(loc.shape, generator=generator, dtype=dtype))[0]
                ns, r, done, _ = env.step(a.numpy().astype(np.float32))
                last = done or t == max_path_length - 1
                S.append(st[0]); A.append(a); Rw.append(r); D.append(1.0 if last else 0.0)
                NS.append(torch.as_tensor(ns, dtype=dtype))
                s = ns
                if done:
                    break
    return dict(states=torch.stack(S), actions=torch.stack(A), rewards=torch.tensor(Rw, dtype=dtype).view(-1, 1),
                dones=torch.tensor(D, dtype=dtype).view(-1, 1), next_states=torch.stack(NS))


# ----------------------------------------------------------------------------------------------- rl.py TRPO part
def compute_advantages(baseline, tau, gamma, ep, update_vf=True):
    """rl.py:95-110"""
    returns = discount(gamma, ep['rewards'], ep['dones'])
    if update_vf:
        baseline.fit(ep['states'], returns)
    values = baseline(ep['states'])
    next_values = baseline(ep['next_states'])
    bootstraps = values * (1.0 - ep['dones']) + next_values * ep['dones']
    return generalized_advantage(gamma, tau, ep['rewards'], ep['dones'], bootstraps, torch.zeros(1, dtype=values.dtype))


def trpo_a2c_loss(ep, p, baseline, gamma, tau, update_vf=True, activation=torch.relu):
    """rl.py:346-358"""
    log_probs = policy_log_prob(p, ep['states'], ep['actions'], activation)
    adv = normalize(compute_advantages(baseline, tau, gamma, ep, update_vf)).detach()
    return a2c_policy_loss(log_probs, adv)


def trpo_update(ep, p, baseline, inner_lr, gamma, tau, first_order=False, activation=torch.relu, head_only=False):
    """rl.py:361-374: grad (create_graph = second order) + learn2learn maml_update (p <- p - lr g).
    head_only restates DiagNormalPolicyANIL with `features_no_grad` (policies.py:100-106): the hidden layers are evaluated
    under no_grad, their gradients are None (allow_unused=anil, rl.py:371) and maml_update leaves them unchanged -- pinned by
    the g5_anil fixtures (gradient with the body switched off)."""
    so = not first_order
    loss = trpo_a2c_loss(ep, p, baseline, gamma, tau, activation=activation)
    grads = torch.autograd.grad(loss, list(p.values()), retain_graph=so, create_graph=so)
    last = max(int(k.split('.')[1]) for k in p if k.startswith('mean.'))
    keep = lambda k: (not head_only) or k == 'sigma' or k.startswith(f'mean.{last}.')
    return OrderedDict((k, v - inner_lr * g if keep(k) else v) for (k, v), g in zip(p.items(), grads))


def fast_adapt_trpo(env, p, baseline, params, generator, first_order=False, activation=torch.relu, anil=False):
    """rl.py:377-406 (rollouts through collect_episodes); anil: the inner updates run with the body grads off (:381-382)."""
    replay = []
    for _ in range(params['adapt_steps']):
        ep = collect_episodes(env, p, params['adapt_batch_size'], params['max_path_length'], generator, activation=activation)
        replay.append(ep)
        p = trpo_update(ep, p, baseline, params['inner_lr'], params['gamma'], params['tau'], first_order=first_order,
                        activation=activation, head_only=anil)
    q = collect_episodes(env, p, params['adapt_batch_size'], params['max_path_length'], generator, activation=activation)
    replay.append(q)
    valid_loss = trpo_a2c_loss(q, p, baseline, params['gamma'], params['tau'], update_vf=False, activation=activation)
    return p, valid_loss, replay, q['rewards'].sum().item() / params['adapt_batch_size']


def get_ep_successes(success, path_length):
    """rl.py:59-72: ``success`` = the replay's per-step success flags; reshape(path_length, -1).T lays one episode per row (the
    reference's runner interleaves its workers' steps), an episode counts when any of its flags is 1."""
    if success is None:                                   # AttributeError branch (rl.py:69-71): 'No success metric registered!'
        return 0
    return int(sum(1 for ep in success.reshape(path_length, -1).T if 1. in ep))


def fast_adapt_trpo_replayed(replays, p, baseline, params, first_order=False, activation=torch.relu, anil=False, success=None):
    """rl.py:377-406 on GIVEN replays (``task.run`` hands out replays[0..K-1] as the support episodes, replays[K] as the query):
    K x trpo_update (head-only under anil: the body grads are off, :381-382), the validation loss WITHOUT refitting the baseline
    (:401), mean query reward (:403), success rate (:404)."""
    for k in range(params['adapt_steps']):
        p = trpo_update(replays[k], p, baseline, params['inner_lr'], params['gamma'], params['tau'], first_order=first_order,
                        activation=activation, head_only=anil)
    q = replays[params['adapt_steps']]
    valid_loss = trpo_a2c_loss(q, p, baseline, params['gamma'], params['tau'], update_vf=False, activation=activation)
    rew = q['rewards'].sum().item() / params['adapt_batch_size']
    suc = get_ep_successes(success, params['max_path_length']) / params['adapt_batch_size']
    return p, valid_loss, rew, suc


def meta_surrogate_loss(iter_replays, iter_policies, p, baseline, params, activation=torch.relu):
    """rl.py:441-473"""
    mean_loss, mean_kl = 0.0, 0.0
    for task_replays, old in zip(iter_replays, iter_policies):
        new = OrderedDict((k, v.clone()) for k, v in p.items())               # clone_module
        for ep in task_replays[:-1]:
            new = trpo_update(ep, new, baseline, params['inner_lr'], params['gamma'], params['tau'], first_order=False,
                              activation=activation)
        v = task_replays[-1]
        old_loc, old_scale = policy_loc_scale(old, v['states'], activation)
        new_loc, new_scale = policy_loc_scale(new, v['states'], activation)
        mean_kl = mean_kl + normal_kl(new_loc, new_scale, old_loc, old_scale).mean()
        adv = normalize(compute_advantages(baseline, params['tau'], params['gamma'], v)).detach()
        old_lp = normal_log_prob(old_loc, old_scale, v['actions']).mean(dim=1, keepdim=True).detach()
        new_lp = normal_log_prob(new_loc, new_scale, v['actions']).mean(dim=1, keepdim=True)
        mean_loss = mean_loss + trpo_policy_loss(new_lp, old_lp, adv)
    return mean_loss / len(iter_replays), mean_kl / len(iter_replays)


def meta_optimize_trpo(params, p, baseline, iter_replays, iter_policies, activation=torch.relu):
    """rl.py:409-438.  ``p``: OrderedDict of leaf tensors (requires_grad); updated in place.  Returns diagnostics."""
    old_loss, old_kl = meta_surrogate_loss(iter_replays, iter_policies, p, baseline, params, activation)
    plist = list(p.values())
    grad = torch.autograd.grad(old_loss, plist, retain_graph=True)
    grad = torch.cat([g.detach().reshape(-1) for g in grad])
    Fvp = hessian_vector_product(old_kl, plist)
    step = conjugate_gradient(Fvp, grad)
    shs = 0.5 * torch.dot(step, Fvp(step))
    lagrange = torch.sqrt(shs / params['max_kl'])
    step = step / lagrange
    old_loss = old_loss.detach()
    accepted, new_loss, kl = None, None, None
    for ls_step in range(params['ls_max_steps']):
        stepsize = params['backtrack_factor'] ** ls_step * params['outer_lr']
        cand, off = OrderedDict(), 0
        for k, v in p.items():
            n = v.numel()
            cand[k] = (v.detach() - stepsize * step[off:off + n].view_as(v)).requires_grad_(True)
            off += n
        new_loss, kl = meta_surrogate_loss(iter_replays, iter_policies, cand, baseline, params, activation)
        if new_loss < old_loss and kl < params['max_kl']:
            with torch.no_grad():
                for k in p:
                    p[k].copy_(cand[k])
            accepted = ls_step
            break
    return dict(grad=grad, step=step, old_loss=old_loss, accepted=accepted, new_loss=new_loss, kl=kl,
                fvp=lambda v: Fvp(v))


# ----------------------------------------------------------------------------------------------- rl.py VPG / PPO part (replayed)
def maml_adapt_policy(loss, p, lr, first_order, head_only=False):
    """learn2learn MAML.adapt (rl.py:241,292,335): g = grad(loss, params, create_graph = second order, allow_unused = anil);
    p <- p - lr g for the parameters that received a gradient (with the ANIL body under no_grad: sigma and the last Linear)."""
    so = not first_order
    last = max(int(k.split('.')[1]) for k in p if k.startswith('mean.'))
    keep = [k for k in p if (not head_only) or k == 'sigma' or k.startswith(f'mean.{last}.')]
    grads = torch.autograd.grad(loss, [p[k] for k in keep], retain_graph=so, create_graph=so)
    new = OrderedDict(p)
    for k, g in zip(keep, grads):
        new[k] = p[k] - lr * g
    return new


def _body_detached(p, head_only):
    """DiagNormalPolicyANIL.forward_pass with features_no_grad (policies.py:100-106): the hidden layers see no gradient."""
    if not head_only:
        return p
    last = max(int(k.split('.')[1]) for k in p if k.startswith('mean.'))
    return OrderedDict((k, v if (k == 'sigma' or k.startswith(f'mean.{last}.')) else v.detach()) for k, v in p.items())


def magic_box(x):
    """learn2learn.magic_box (rl.py:5,225): exp(x - stop_gradient(x)) -- evaluates to 1, differentiates like x."""
    return torch.exp(x - x.detach())


def dice_log_probs(log_probs, dones):
    """rl.py:219-225 (vpg_a2c_loss, dice=True):
        weights = ones_like(dones); weights[1:] -= dones[:-1]; weights /= dones.sum()
        cum = weighted_cumsum(log_probs, weights)        # rl.py:202-205: for i in range(N): values[i] += values[i-1] * weights[i]
        log_probs = magic_box(cum)
    The loop is in place and starts at i = 0, where values[i-1] is Python's values[-1]: the LAST log-prob, still unmodified.
    Restated out of place (autograd-friendly), same arithmetic."""
    weights = torch.ones_like(dones)
    weights[1:] = weights[1:] - dones[:-1]
    weights = weights / dones.sum()
    n = log_probs.shape[0]
    cum = [log_probs[0] + log_probs[n - 1] * weights[0]]
    for i in range(1, n):
        cum.append(log_probs[i] + cum[i - 1] * weights[i])
    return magic_box(torch.stack(cum))


def vpg_a2c_loss(ep, p, baseline, gamma, tau, dice=False, activation=torch.relu, anil=False):
    """rl.py:208-228 (advantages are not normalised here)."""
    adv = compute_advantages(baseline, tau, gamma, ep).detach()
    lp = policy_log_prob(_body_detached(p, anil), ep['states'], ep['actions'], activation)
    if dice:
        lp = dice_log_probs(lp, ep['dones'])
    return a2c_policy_loss(lp, adv)


def replay_vpg(p, support, query, params, baseline, first_order=False, activation=torch.relu, anil=False, dice=False):
    """fast_adapt_vpg (rl.py:231-255) on given replays: one a2c update per support replay, validation loss = vpg_a2c_loss.
    dice: every vpg_a2c_loss with dice=True (the reference's call sites leave the default False)."""
    for ep in support:
        loss = vpg_a2c_loss(ep, p, baseline, params['gamma'], params['tau'], dice, activation, anil)
        p = maml_adapt_policy(loss, p, params['inner_lr'], first_order, head_only=anil)
    return vpg_a2c_loss(query, p, baseline, params['gamma'], params['tau'], dice, activation), p


def replay_ppo(p, support, query, params, baseline, activation=torch.relu, anil=False):
    """fast_adapt_ppo (rl.py:267-318) on given replays: ppo_epochs clipped-surrogate updates per support replay (second order:
    the reference calls learner.adapt without first_order), validation loss = ppo loss against the adapted policy itself."""
    for ep in support:
        adv = normalize(compute_advantages(baseline, params['tau'], params['gamma'], ep)).detach()
        with torch.no_grad():
            old = policy_log_prob(p, ep['states'], ep['actions'], activation)
        for _ in range(params['ppo_epochs']):
            new = policy_log_prob(_body_detached(p, anil), ep['states'], ep['actions'], activation)
            p = maml_adapt_policy(ppo_policy_loss(new, old, adv, params['ppo_clip_ratio']), p, params['inner_lr'], False, head_only=anil)
    adv = normalize(compute_advantages(baseline, params['tau'], params['gamma'], query)).detach()
    with torch.no_grad():
        old = policy_log_prob(p, query['states'], query['actions'], activation)
    new = policy_log_prob(p, query['states'], query['actions'], activation)
    return ppo_policy_loss(new, old, adv, params['ppo_clip_ratio']), p
