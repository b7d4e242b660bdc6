#!/usr/bin/env python3
"""Generate tests/golden/golden_rl.npz by running the REFERENCE's own TRPO composition (core_functions/rl.py).

Run only in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_rl.py

What is reference code and what is not (same recipe as make_golden.py, which pins the vision half):
  * imported unmodified from /root/reference and EXECUTED: ``core_functions.rl.{compute_advantages (:95-110), trpo_a2c_loss
    (:346-358), trpo_update (:361-374), fast_adapt_trpo (:377-406), meta_optimize_trpo (:409-438), meta_surrogate_loss
    (:441-473), get_episode_values (:49-56), get_ep_successes (:59-72)}`` and ``core_functions.policies.{DiagNormalPolicy,
    DiagNormalPolicyANIL}``;
  * the third-party LEAVES those lines call -- cherry's ``td.discount``, ``pg.generalized_advantage``, ``normalize``,
    ``a2c.policy_loss``, ``trpo.policy_loss``, ``trpo.hessian_vector_product``, ``trpo.conjugate_gradient``,
    ``LinearValue``; learn2learn's ``clone_module`` / ``maml_update`` -- are absent from /root/reference (un-vendored,
    unpinned).  The restatements of ``oracle/rl_ref.py`` are installed in their place (module attributes of the imported
    reference module), exactly as ``StandInLearner`` stands in for l2l in the vision fixtures.  So these fixtures pin the
    oracle's COMPOSITION (which leaf is called with what, in which order, detached where, averaged how) to the reference's
    lines; the leaves themselves stay "parity unpinned" (oracle/__init__.py);
  * replays are duck types with cherry ExperienceReplay's five accessors (``state() action() reward() done() next_state()``,
    rl.py:49-56) over tensors collected with the oracle's Particles2D stand-in; the runner handed to the reference's
    ``fast_adapt_trpo`` returns pre-collected replays in order (the reference's own Runner needs gym / cherry).
Inputs are stored next to the outputs (small cases) or as checksums of the seeded generator's output (cfg5 size).
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.dont_write_bytecode = True

from oracle import rl_ref as RL  # noqa: E402
from rl_cases import CASES, make_case, FIELDS  # noqa: E402  (tests/rl_cases.py: the seeded inputs, shared with the tests)
import make_golden  # noqa: E402  (the stub recipe)


# ------------------------------------------------------------------------------------------ learn2learn stand-ins (module form)
def clone_module(module):
    """learn2learn.clone_module (published semantics): a new module object of the same class whose parameters are
    ``param.clone()`` -- differentiable copies with a graph edge to the originals -- recursively; buffers shared."""
    if not isinstance(module, torch.nn.Module):
        return module
    clone = module.__new__(type(module))
    clone.__dict__ = module.__dict__.copy()
    clone._parameters = {k: (None if p is None else p.clone()) for k, p in module._parameters.items()}
    clone._buffers = dict(module._buffers)
    clone._modules = {k: clone_module(m) for k, m in module._modules.items()}
    return clone


def maml_update(model, lr, grads=None):
    """learn2learn.algorithms.maml.maml_update (published semantics): p <- p + (-lr * g) out of place for every parameter that
    received a gradient (``g is None`` -- allow_unused -- leaves it), written back into the module; returns the module."""
    params = list(model.parameters())
    assert len(params) == len(list(grads))
    new = {id(p): p + (-lr * g) for p, g in zip(params, grads) if g is not None}

    def walk(m):
        for k, p in m._parameters.items():
            if p is not None and id(p) in new:
                m._parameters[k] = new[id(p)]
        for sub in m._modules.values():
            walk(sub)
    walk(model)
    return model


class Episodes:
    """Duck type of cherry's ExperienceReplay as rl.py:49-56 reads it (+ ``success()`` when given, rl.py:59-72)."""

    def __init__(self, d, success=None):
        self._d, self._success = d, success
        if success is not None:
            self.success = lambda: self._success

    def state(self): return self._d['states']
    def action(self): return self._d['actions']
    def reward(self): return self._d['rewards']
    def done(self): return self._d['dones']
    def next_state(self): return self._d['next_states']


class ReplayRunner:
    """Stands in for core_functions/runner.py::Runner in the reference's fast_adapt_trpo: ``run`` hands out pre-collected replays."""

    def __init__(self, replays):
        self.replays, self.i = list(replays), 0

    def run(self, learner, episodes=None, render=False):
        r = self.replays[self.i]
        self.i += 1
        return r


def import_reference_rl():
    make_golden.import_reference()                         # stubs + sys.path
    import core_functions.rl as rl
    ns = types.SimpleNamespace
    rl.ch.td = ns(discount=RL.discount)
    rl.ch.normalize = RL.normalize
    rl.generalized_advantage = lambda tau, gamma, rewards, dones, values, next_value: \
        RL.generalized_advantage(gamma, tau, rewards, dones, values, next_value)
    rl.a2c = ns(policy_loss=RL.a2c_policy_loss)
    rl.trpo = ns(policy_loss=RL.trpo_policy_loss, hessian_vector_product=RL.hessian_vector_product,
                 conjugate_gradient=RL.conjugate_gradient)
    rl.clone_module = clone_module
    rl.maml_update = maml_update
    rl.set_device(torch.device('cpu'))
    from core_functions.policies import DiagNormalPolicy, DiagNormalPolicyANIL
    return rl, DiagNormalPolicy, DiagNormalPolicyANIL


def ref_policy(cls, theta, dt, anil=False):
    pol = cls(2, 2, 100) if anil else cls(2, 2)
    pol.to(dt)
    names = [k for k, _ in pol.named_parameters()]
    src = theta
    if anil:                                               # mean.{0,2}.* -> body.{0,2}.*, mean.4.* -> head.*
        src = {}
        for k, v in theta.items():
            if k.startswith('mean.4.'):
                src['head.' + k[len('mean.4.'):]] = v
            elif k.startswith('mean.'):
                src['body.' + k[len('mean.'):]] = v
            else:
                src[k] = v
    assert set(names) == set(src), (names, list(src))
    with torch.no_grad():
        for k, p in pol.named_parameters():
            p.copy_(src[k].to(dt))
    return pol


def flat(params):
    return torch.cat([p.detach().reshape(-1) for p in params])


def flat_engine_order(pol):
    """sigma, then the Linear layers in forward order: the order of oracle/rl_ref.py's parameter dicts (DiagNormalPolicy's own
    named_parameters() order; for the ANIL class sigma, body.*, head.* -- the same sequence)."""
    return flat([p for _, p in pol.named_parameters()])


def run_case(rl, classes, name, final):
    spec = CASES[name]
    anil = spec.get('anil', False)
    cls = classes[1] if anil else classes[0]
    rec64 = {}
    for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        out = rec64 if tag == 'f64' else {}
        case = make_case(name, dt)
        params, theta, replays, olds = case['params'], case['theta'], case['replays'], case['olds']
        T = len(replays)
        pre = f'rl_{name}_{tag}'
        eps = [[Episodes(r) for r in task] for task in replays]

        # --- rl.py:95-110 compute_advantages (update_vf True, then False on the other replay with the weights just fitted)
        bl = RL.LinearValue(2, 2)
        r0, r1 = replays[0][0], replays[0][-1]
        a_fit = rl.compute_advantages(bl, params['tau'], params['gamma'], r0['rewards'], r0['dones'], r0['states'], r0['next_states'])
        a_nofit = rl.compute_advantages(bl, params['tau'], params['gamma'], r1['rewards'], r1['dones'], r1['states'], r1['next_states'],
                                        update_vf=False)
        out[f'{pre}_adv_fit'] = a_fit.numpy()
        out[f'{pre}_adv_nofit'] = a_nofit.numpy()
        out[f'{pre}_vf_weight'] = bl.weight.numpy()

        # --- rl.py:346-358 trpo_a2c_loss and its gradient; rl.py:361-374 trpo_update (first and second order give the same values)
        pol = ref_policy(cls, theta, dt, anil)
        loss = rl.trpo_a2c_loss(eps[0][0], pol, RL.LinearValue(2, 2), params['gamma'], params['tau'])
        g = torch.autograd.grad(loss, list(pol.parameters()))
        out[f'{pre}_inner_loss'] = np.array([loss.item()])
        out[f'{pre}_inner_grad'] = flat(g).numpy()
        new = rl.trpo_update(eps[0][0], clone_module(pol), RL.LinearValue(2, 2), params['inner_lr'], params['gamma'], params['tau'],
                             anil=anil, first_order=True)
        if tag == 'f64':
            out[f'{pre}_adapted_theta'] = flat(new.parameters()).numpy()

        # --- rl.py:377-406 fast_adapt_trpo on the stored replays (the runner replays them): adapted parameters, the validation loss with
        # update_vf=False, mean query reward, success rate (with and without a success() accessor)
        learner = copy.deepcopy(pol)
        if anil:
            object.__setattr__(learner, 'module', learner)  # rl.py:382,396 reach the policy through the l2l wrapper's .module
        L = params['max_path_length']
        q = replays[0][-1]
        full = q['states'].shape[0] == L * params['adapt_batch_size']
        succ = None
        if full:                                            # rl.py:64 reshape(path_length, -1).T needs full-length episodes
            n_q = q['states'].shape[0]                       # a sparse deterministic pattern: some episodes succeed, some never do
            succ = (torch.arange(n_q) % (n_q // 3 + 3) == 0).to(dt)
        task_eps = [Episodes(r) for r in replays[0][:-1]] + [Episodes(q, success=succ)]
        bl = RL.LinearValue(2, 2)
        adapted, vloss, rep, rew, suc = rl.fast_adapt_trpo(ReplayRunner(task_eps), learner, bl, params, anil=anil, first_order=True)
        if tag == 'f64':
            out[f'{pre}_fa_theta'] = flat(adapted.parameters()).numpy()
        out[f'{pre}_fa_valid_loss'] = np.array([vloss.item()])
        out[f'{pre}_fa_reward_success'] = np.array([rew, suc])
        out[f'{pre}_fa_has_success'] = np.array([int(full)])
        if succ is not None:
            out[f'{pre}_fa_success_flags'] = succ.numpy()

        # --- rl.py:441-473 meta_surrogate_loss at theta; gradient of the loss (:413-416)
        pol = ref_policy(cls, theta, dt, anil)
        old_pols = [ref_policy(cls, o, dt, anil) for o in olds]
        sl, kl = rl.meta_surrogate_loss(eps, old_pols, pol, RL.LinearValue(2, 2), params, anil)
        sg = torch.autograd.grad(sl, list(pol.parameters()), retain_graph=True)
        out[f'{pre}_surr_loss_kl'] = np.array([sl.item(), kl.item()])
        out[f'{pre}_surr_grad'] = flat(sg).numpy()
        # a displaced candidate (what the line search evaluates)
        cand = copy.deepcopy(pol)
        with torch.no_grad():
            for p in cand.parameters():
                p.add_(0.01 * torch.sin(torch.arange(p.numel(), dtype=dt)).view_as(p))
        sl2, kl2 = rl.meta_surrogate_loss(eps, old_pols, cand, RL.LinearValue(2, 2), params, anil)
        out[f'{pre}_surr_displaced_loss_kl'] = np.array([sl2.item(), kl2.item()])

        # --- rl.py:409-438 meta_optimize_trpo: record every Fisher-vector product and every line-search evaluation by wrapping
        # the reference module's OWN callees (the wrapped functions are still the ones that run)
        fvp_in, fvp_out, evals = [], [], []
        hvp0, msl0 = rl.trpo.hessian_vector_product, rl.meta_surrogate_loss

        def hvp_rec(loss_, ps_):
            f = hvp0(loss_, ps_)
            def call(v):
                r = f(v)
                fvp_in.append(v.detach().clone())
                fvp_out.append(r.detach().clone())
                return r
            return call

        def msl_rec(*a, **k):
            l_, k_ = msl0(*a, **k)
            evals.append((l_.item(), k_.item()))
            return l_, k_
        rl.trpo.hessian_vector_product, rl.meta_surrogate_loss = hvp_rec, msl_rec
        try:
            pol = ref_policy(cls, theta, dt, anil)
            before = flat(pol.parameters()).clone()
            rl.meta_optimize_trpo(params, pol, RL.LinearValue(2, 2), eps, old_pols, anil=anil)
        finally:
            rl.trpo.hessian_vector_product, rl.meta_surrogate_loss = hvp0, msl0
        after = flat(pol.parameters())
        moved = not torch.equal(before, after)
        out[f'{pre}_opt_evals'] = np.array(evals)                              # [1 + line-search evaluations, (loss, kl)]
        out[f'{pre}_opt_accepted'] = np.array([len(evals) - 2 if moved else -1])
        out[f'{pre}_opt_theta_new'] = after.numpy()
        out[f'{pre}_opt_n_fvp'] = np.array([len(fvp_out)])
        out[f'{pre}_opt_fvp_dots'] = np.array([[torch.dot(a, b).item(), b.norm().item()] for a, b in zip(fvp_in, fvp_out)])
        if tag == 'f64':                                                       # (the fp32 run: scalars, gradient, step and new parameters only)
            out[f'{pre}_opt_fvp_first'] = fvp_out[0].numpy()                   # F g  (the first CG product: p = g)
            out[f'{pre}_opt_fvp_last'] = fvp_out[-1].numpy()                   # F step (rl.py:419)
        out[f'{pre}_opt_cg_step'] = fvp_in[-1].numpy()                         # the unscaled CG solution
        print(name, tag, 'surr', out[f'{pre}_surr_loss_kl'], 'evals', len(evals), 'accepted', out[f'{pre}_opt_accepted'],
              'n_fvp', len(fvp_out), flush=True)

        if tag == 'f32':
            # the reference's own fp32 run: scalars, and how far its vectors sit from its fp64 run (the calibration SURVEY.md 8c made for the
            # vision path: what "fp32 parity" can mean on this path).  Not bit-reproducible run to run (the fp32 normal equations of the
            # baseline fit amplify the BLAS's summation order), so nothing asserts on these beyond their order of magnitude.
            rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))
            for k in ('inner_loss', 'fa_valid_loss', 'fa_reward_success', 'surr_loss_kl', 'surr_displaced_loss_kl', 'opt_evals', 'opt_accepted'):
                final[f'{pre}_{k}'] = out[f'{pre}_{k}']
            final[f'{pre}_rel_to_f64'] = np.array([rel(out[f'{pre}_{k}'], rec64[f'rl_{name}_f64_{k}'])
                                                   for k in ('adv_fit', 'inner_grad', 'surr_grad', 'opt_cg_step', 'opt_theta_new')])
            continue
        # inputs: stored in full for the small cases, as checksums always
        chk = []
        for task in replays:
            for r in task:
                chk.append([float(r[k].double().sum()) for k in FIELDS] + [float(r['states'].shape[0])])
        out[f'{pre}_input_checksums'] = np.array(chk)
        if spec.get('store_inputs', False) and tag == 'f64':
            for t, task in enumerate(replays):
                for j, r in enumerate(task):
                    for k in FIELDS:
                        out[f'rl_{name}_in_t{t}_r{j}_{k}'] = r[k].numpy()
            for t, o in enumerate(olds):
                out[f'rl_{name}_in_old{t}'] = torch.cat([v.reshape(-1) for v in o.values()]).numpy()
            out[f'rl_{name}_in_theta'] = torch.cat([v.reshape(-1) for v in theta.values()]).numpy()
    final.update(rec64)


def main():
    torch.set_num_threads(1)                               # one thread: every reduction order fixed, the fixture regenerates bit for bit
    rl, pol_cls, anil_cls = import_reference_rl()
    out = {}
    for name in CASES:
        run_case(rl, (pol_cls, anil_cls), name, out)
    path = os.path.join(HERE, 'golden_rl.npz')
    np.savez_compressed(path, **out)
    print('golden_rl.npz', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
