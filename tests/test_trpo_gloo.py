"""N>1 path of MAML-TRPO on CPU: world_size-2 gloo run of core_functions.rl.meta_optimize_trpo with a RAGGED task split
(3 tasks: 2 + 1) must reproduce the single-process run over all tasks -- the surrogate loss / KL / gradient, every
Fisher-vector product inside conjugate gradient, the accepted line-search index and the updated parameters.  The collective
plumbing under test (rl._SurrogateContext._allmean: task-count-weighted means, 1 + 11 + <=15 all-reduces per iteration,
reference core_functions/rl.py:409-473) is the product's; the per-rank compute is injected: a CPU stand-in for PolicyEngine
written with autograd on the oracle's policy functions (no GPU in this container)."""
import os
import socket
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PARAMS = dict(inner_lr=0.1, max_path_length=12, adapt_steps=1, adapt_batch_size=4, meta_batch_size=3, outer_lr=0.3,
              backtrack_factor=0.5, ls_max_steps=15, max_kl=0.01, tau=1.0, gamma=0.99)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _CpuPolicyEngine:
    """The calls _SurrogateContext makes on PolicyEngine (forward / surrogate / fvp), restated with autograd in fp64 on padded
    [T, B, *] batches with per-task row counts."""

    def __init__(self, RL, shapes):
        self.RL, self.shapes = RL, shapes

    def _named(self, flat):
        out, off = OrderedDict(), 0
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            out[k] = flat[off:off + n].view(shp)
            off += n
        return out

    def forward(self, theta, states):
        locs = []
        for t in range(states.shape[0]):
            th = theta if theta.dim() == 1 else theta[t]
            locs.append(self.RL.policy_loc_scale(self._named(th.double()), states[t].double())[0])
        return torch.stack(locs).float()

    def _loss_kl(self, theta, sup, qry, old_loc, old_scale, inner_lr):
        RL = self.RL
        loss, kl = 0.0, 0.0
        T = qry['states'].shape[0]
        for t in range(T):
            p = self._named(theta)
            ns, nq = int(sup['count'][t]), int(qry['count'][t])
            lp = RL.policy_log_prob(p, sup['states'][t, :ns].double(), sup['actions'][t, :ns].double())
            inner = RL.a2c_policy_loss(lp, sup['adv'][t, :ns].double().view(-1, 1))
            g = torch.autograd.grad(inner, list(p.values()), create_graph=True)
            new = OrderedDict((k, v - inner_lr * gi) for (k, v), gi in zip(p.items(), g))
            loc, scale = RL.policy_loc_scale(new, qry['states'][t, :nq].double())
            ol, osc = old_loc[t, :nq].double(), old_scale[t].double()
            kl = kl + RL.normal_kl(loc, scale, ol, osc).mean()
            new_lp = RL.normal_log_prob(loc, scale, qry['actions'][t, :nq].double()).mean(dim=1, keepdim=True)
            old_lp = RL.normal_log_prob(ol, osc, qry['actions'][t, :nq].double()).mean(dim=1, keepdim=True)
            loss = loss + RL.trpo_policy_loss(new_lp, old_lp, qry['adv'][t, :nq].double().view(-1, 1))
        return loss / T, kl / T

    def surrogate(self, theta, sup, qry, old_loc, old_scale, inner_lr, want_grad):
        th = theta.double().clone().requires_grad_(True)
        loss, kl = self._loss_kl(th, sup, qry, old_loc, old_scale, inner_lr)
        grad = torch.autograd.grad(loss, th)[0].float() if want_grad else None
        self._last = (sup, qry, old_loc, old_scale, inner_lr)
        return loss.detach().float().view(1), kl.detach().float().view(1), grad

    def fvp(self, theta, sup, qry, inner_lr, damping, v):
        _, _, old_loc, old_scale, _ = self._last
        th = theta.double().clone().requires_grad_(True)
        _, kl = self._loss_kl(th, sup, qry, old_loc, old_scale, inner_lr)
        g = torch.autograd.grad(kl, th, create_graph=True)[0]
        h = torch.autograd.grad(torch.dot(g, v.double()), th)[0]
        return (h + damping * v.double()).float()


class _StubPolicy:
    """What _SurrogateContext / meta_optimize_trpo touch on a policy object."""

    def __init__(self, flat, engine, shapes):
        self._flat, self._engine = flat.clone(), engine
        self.input_size, self.output_size = 2, 2
        n_sigma = int(np.prod(shapes['sigma']))
        self._n_sigma = n_sigma

    @property
    def sigma(self):
        return self._flat[:self._n_sigma]              # 'sigma' is the first registered parameter (policies.py:30-47)

    def engine(self):
        return self._engine

    def flat(self):
        return self._flat.clone()

    def load_flat(self, theta):
        self._flat = theta.detach().clone()


def _run(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch.distributed as dist
    from exploring_meta_amd.core_functions import rl as prl
    from exploring_meta_amd.sharding import shard_range
    from oracle import rl_ref as RL
    from helpers import hash_params
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    shapes = RL.policy_param_shapes()
    theta = hash_params(shapes, 19)
    theta['sigma'] = torch.tensor([-0.3, 0.2], dtype=torch.float64)
    env, gen, baseline = RL.Particles2D(seed=1), torch.Generator().manual_seed(2), RL.LinearValue(2, 2)
    replays, olds = [], []
    for task in env.sample_tasks(PARAMS['meta_batch_size']):            # every rank generates the same global task list ...
        env.set_task(task)
        learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
        adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, baseline, PARAMS, gen, first_order=True)
        replays.append([{k: v.float() for k, v in ep.items()} for ep in rep])
        olds.append(torch.cat([v.detach().reshape(-1) for v in adapted.values()]).float())
    a, b = shard_range(len(replays), rank, world)                        # ... and keeps its contiguous shard (2 + 1 tasks)
    eng = _CpuPolicyEngine(RL, shapes)
    flat0 = torch.cat([v.reshape(-1) for v in theta.values()]).float()
    pol = _StubPolicy(flat0, eng, shapes)
    old_pols = [_StubPolicy(o, eng, shapes) for o in olds[a:b]]
    prl.set_device(torch.device('cpu'))
    out = prl.meta_optimize_trpo(PARAMS, pol, prl.LinearValue(2, 2), replays[a:b], old_pols)
    v = torch.sin(torch.arange(flat0.numel(), dtype=torch.float32))
    fv = out['fvp'](v)
    if world > 1:
        gathered = [torch.zeros_like(flat0) for _ in range(world)]
        dist.all_gather(gathered, pol.flat())
        assert all(torch.equal(g, gathered[0]) for g in gathered)        # identical parameters on every rank
        dist.destroy_process_group()
    if rank == 0:
        torch.save(dict(theta=pol.flat(), grad=out['grad'], step=out['step'], accepted=out['accepted'], fv=fv,
                        old_loss=float(out['old_loss']), old_kl=float(out['old_kl']), new_loss=float(out['new_loss']),
                        kl=float(out['kl'])), out_path)


def test_trpo_two_ranks_ragged_match_single_process(tmp_path):
    single, multi = str(tmp_path / 'single.pt'), str(tmp_path / 'multi.pt')
    _run(0, 1, 0, single)
    mp.spawn(_run, args=(2, _free_port(), multi), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(multi)
    assert a['accepted'] is not None and a['accepted'] == b['accepted']
    rel = lambda x, y: float((x - y).norm() / y.norm())
    assert rel(b['grad'], a['grad']) < 1e-5
    assert rel(b['fv'], a['fv']) < 1e-5
    assert rel(b['step'], a['step']) < 1e-3                              # ten CG iterations on fp32 products
    assert rel(b['theta'], a['theta']) < 1e-4
    for k in ('old_loss', 'new_loss'):
        assert a[k] == pytest.approx(b[k], rel=1e-5, abs=1e-6)
    assert abs(a['old_kl'] - b['old_kl']) < 1e-7 and abs(a['kl'] - b['kl']) < 1e-5
