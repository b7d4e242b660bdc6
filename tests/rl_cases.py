"""Seeded inputs of the RL golden cases (tests/golden/golden_rl.npz): shared by the generator (tests/golden/make_golden_rl.py, which
runs the REFERENCE's rl.py on them) and by the tests that hold oracle/rl_ref.py -- and through it the HIP path -- to the recorded
results.  Everything here is oracle-side (CPU, test infrastructure)."""
from collections import OrderedDict

import numpy as np
import torch

from oracle import rl_ref as RL
from helpers import hash_params

FIELDS = ('states', 'actions', 'rewards', 'dones', 'next_states')

PARAMS = dict(inner_lr=0.1, max_path_length=25, adapt_steps=1, adapt_batch_size=6, meta_batch_size=4, outer_lr=0.3,
              backtrack_factor=0.5, ls_max_steps=15, max_kl=0.01, tau=1.0, gamma=0.99)

CASES = OrderedDict(
    # MAML-TRPO as rl/maml_trpo.py runs it, small: 4 tasks x 6 episodes x <= 25 steps (ragged: episodes end early)
    small_relu=dict(params=PARAMS, activation='relu', store_inputs=True),
    # two inner steps (params['adapt_steps'] = 2): the replayed second-order chain of rl.py:450-453
    two_steps=dict(params=dict(PARAMS, adapt_steps=2, inner_lr=0.05), activation='relu', store_inputs=True),
    # ANIL-TRPO (rl/anil_trpo.py): DiagNormalPolicyANIL (tanh body), the stored policies adapted head-only, the surrogate's inner step with all
    # parameters (rl.py:381-382,395-396,447-453); inner_lr of rl/anil_trpo.py:22
    anil_tanh=dict(params=dict(PARAMS, inner_lr=0.01), activation='tanh', anil=True, store_inputs=True),
    # BASELINE config 5 as benchmarked: 20 tasks x 20 episodes x 100 steps (rl/maml_trpo.py:21-33 defaults)
    cfg5=dict(params=dict(PARAMS, max_path_length=100, adapt_batch_size=20, meta_batch_size=20), activation='relu'),
)


def theta64():
    p = hash_params(RL.policy_param_shapes(), 19)
    p['sigma'] = torch.tensor([-0.3, 0.2], dtype=torch.float64)
    return p


def make_case(name, dtype=torch.float64):
    """-> dict(params, theta, replays [task][replay] of field dicts, olds [task] parameter dicts, activation, anil).  Collected in fp64 with the
    oracle's Particles2D stand-in and torch generators (deterministic on one torch build: the fixture's input checksums say whether
    this build reproduces the inputs the fixture was made from), then cast to ``dtype``."""
    spec = CASES[name]
    params = spec['params']
    act = torch.tanh if spec['activation'] == 'tanh' else torch.relu
    anil = spec.get('anil', False)
    env = RL.Particles2D(seed=1)
    gen = torch.Generator().manual_seed(2)
    theta = theta64()
    baseline = RL.LinearValue(2, 2)
    replays, olds = [], []
    for task in env.sample_tasks(params['meta_batch_size']):
        env.set_task(task)
        learner = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in theta.items())
        adapted, _, rep, _ = RL.fast_adapt_trpo(env, learner, baseline, params, gen, first_order=True, activation=act, anil=anil)
        replays.append([{k: r[k].to(dtype) for k in FIELDS} for r in rep])
        olds.append(OrderedDict((k, v.detach().to(dtype)) for k, v in adapted.items()))
    theta = OrderedDict((k, v.to(dtype)) for k, v in theta.items())
    return dict(params=params, theta=theta, replays=replays, olds=olds, activation=act, anil=anil)


def input_checksums(replays):
    return np.array([[float(r[k].double().sum()) for k in FIELDS] + [float(r['states'].shape[0])] for task in replays for r in task])


def load_case(golden, name, dtype=torch.float64):
    """The case's inputs as the fixture recorded them: stored arrays (small cases, exact) or the seeded generator's output checked
    against the recorded checksums (config-5 size)."""
    spec = CASES[name]
    if not spec.get('store_inputs', False):
        case = make_case(name, dtype)
        if dtype == torch.float64:
            want = golden[f'rl_{name}_f64_input_checksums']
            got = input_checksums(case['replays'])
            assert got.shape == want.shape and np.allclose(got, want, rtol=1e-10, atol=1e-9), \
                'this torch build does not reproduce the replays golden_rl.npz was recorded on'
        return case
    params = spec['params']
    names = list(RL.policy_param_shapes().keys())
    shapes = RL.policy_param_shapes()

    def unflat(v):
        out, off = OrderedDict(), 0
        for k in names:
            n = int(np.prod(shapes[k]))
            out[k] = torch.from_numpy(v[off:off + n].reshape(shapes[k]).copy()).to(dtype)
            off += n
        return out
    T, nrep = params['meta_batch_size'], params['adapt_steps'] + 1
    replays = [[{k: torch.from_numpy(golden[f'rl_{name}_in_t{t}_r{j}_{k}']).to(dtype) for k in FIELDS} for j in range(nrep)] for t in range(T)]
    olds = [unflat(golden[f'rl_{name}_in_old{t}']) for t in range(T)]
    return dict(params=params, theta=unflat(golden[f'rl_{name}_in_theta']), replays=replays, olds=olds,
                activation=torch.tanh if spec['activation'] == 'tanh' else torch.relu, anil=spec.get('anil', False))


def success_flags(n, dtype=torch.float64):
    """The sparse success pattern the generator attached to the query replay of task 0 (rl.py:59-72 reads ``episodes.success()``)."""
    return (torch.arange(n) % (n // 3 + 3) == 0).to(dtype)
