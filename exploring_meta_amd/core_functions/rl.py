"""MAML-TRPO functions with the reference's names (core_functions/rl.py TRPO part, lines 95-110 and 346-473) on the batched
HIP policy engine.  Replays are either dicts of tensors ``states [N,S], actions [N,A], rewards [N,1], dones [N,1], next_states
[N,S]`` (episodes concatenated; what this package's runners return) or any object with cherry ExperienceReplay's accessors
``state() action() reward() done() next_state()`` (+ ``success()`` where the runner records it) -- what the reference's own call
sites pass (rl.py:49-56, rl/maml_trpo.py:106-134); ``_as_replay`` reads either.
Host side (tiny): discounting, the LinearValue least-squares baseline, GAE, normalisation.
Device side: policy forward, the inner ``trpo_update``, the meta surrogate loss / KL, its gradient, and the Fisher-vector
products inside conjugate gradient -- for ALL tasks of the meta-batch per call.
"""
from copy import deepcopy

import numpy as np
import torch

device = torch.device('cuda')


def set_device(dev):
    """reference rl.py:44-46"""
    global device
    device = dev


# ---------------------------------------------------------------------------------------------- replays
_ACCESSORS = (('states', 'state'), ('actions', 'action'), ('rewards', 'reward'), ('dones', 'done'), ('next_states', 'next_state'))


def _as_replay(episodes):
    """A replay as the dict the rest of this module reads.  Dicts pass through; an object with cherry ExperienceReplay's accessors
    (reference rl.py:49-56 ``get_episode_values``: ``state() action() reward() done() next_state()``) is read ONCE into a ``Replay``
    that is remembered on the object (``_mi_replay``), so the meta-step that walks the same replays again (rl.py:444-465) finds the same
    tensors -- and the device addresses already checked for them.  ``success()`` (rl.py:59-72), where present, rides along."""
    if isinstance(episodes, dict):
        return episodes
    memo = getattr(episodes, '_mi_replay', None)
    if memo is not None:
        return memo
    try:
        fields = {k: getattr(episodes, a)() for k, a in _ACCESSORS}
    except AttributeError as e:
        raise TypeError(f'a replay must be a dict of tensors or offer state() / action() / reward() / done() / next_state(): {e}') from None
    n = fields['states'].shape[0]
    fields['rewards'], fields['dones'] = fields['rewards'].reshape(n, 1), fields['dones'].reshape(n, 1)
    suc = getattr(episodes, 'success', None)
    if callable(suc):
        fields['success'] = suc()
    out = Replay(fields)
    try:
        episodes._mi_replay = out
    except Exception:                                      # (objects without a __dict__: converted again next time)
        pass
    return out


_warned_no_success = [False]


def get_ep_successes(episodes, path_length):
    """reference rl.py:59-72: the number of episodes of the replay with at least one success flag; ``success`` reshaped
    (path_length, -1).T puts one episode per row (the reference's runner interleaves its workers' steps).  Without a success
    record the reference prints 'No success metric registered!' and counts 0 -- here the notice is printed once per process."""
    suc = _as_replay(episodes).get('success')
    if suc is None:
        if not _warned_no_success[0]:
            _warned_no_success[0] = True
            print('No success metric registered!')
        return 0
    suc = torch.as_tensor(suc).detach().reshape(path_length, -1).T
    return int((suc == 1.).any(dim=1).sum().item())


# ---------------------------------------------------------------------------------------------- host-side pieces (cherry semantics)
def _np(x):
    return x.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(x) else np.asarray(x, dtype=np.float64)


def discount(gamma, rewards, dones):
    """cherry.td.discount: R_t = r_t + gamma (1 - d_t) R_{t+1}.  A `done` cuts the recursion, so the replay splits into episodes;
    they are padded with zeros at the END to a common length (zeros behind the last step leave the backward recursion at exactly
    0) and the whole replay is one linear-filter call along the time axis -- bit-identical to the step-by-step loop."""
    try:
        from scipy.signal import lfilter
    except ImportError:                                    # scipy is optional (not in the reference's requirements): plain recursion
        lfilter = None
    r = rewards[:, 0]
    n = r.shape[0]
    ends = np.flatnonzero(dones[:, 0] != 0.0)
    if ends.size == 0 or ends[-1] != n - 1:
        ends = np.append(ends, n - 1)
    starts = np.concatenate([[0], ends[:-1] + 1])
    lens = ends - starts + 1
    L = int(lens.max())
    ep = np.repeat(np.arange(lens.size), lens)             # episode of every step
    pos = np.arange(n) - starts[ep]                        # position inside its episode
    pad = np.zeros((lens.size, L))
    pad[ep, pos] = r
    if lfilter is not None:
        disc = lfilter([1.0], [1.0, -gamma], pad[:, ::-1], axis=1)[:, ::-1]
    else:
        disc = pad.copy()
        for i in range(L - 2, -1, -1):
            disc[:, i] = pad[:, i] + gamma * disc[:, i + 1]
    out = np.empty_like(rewards)
    out[:, 0] = disc[ep, pos]
    return out


class LinearValue:
    """cherry.models.robotics.LinearValue(input_size, reg): features [s, s^2, t, t^2, t^3, 1], t = arange(N)/100; ridge
    normal equations solved by least squares (rl/maml_trpo.py:85 passes env.action_size as ``reg``)."""

    def __init__(self, input_size, reg=1e-5):
        self.input_size, self.reg = input_size, reg
        self._weight, self._weight_dev = np.zeros((2 * input_size + 4, 1)), None

    @property
    def weight(self):
        """The last fitted weights.  A fit done on the GPU (mi_gae_advantages) leaves them on the device; they are fetched on demand."""
        if self._weight_dev is not None:
            self._weight, self._weight_dev = self._weight_dev.detach().cpu().numpy().astype(np.float64).reshape(-1, 1), None
        return self._weight

    @weight.setter
    def weight(self, value):
        self._weight, self._weight_dev = value, None

    def _features(self, states):
        n = states.shape[0]
        al = (np.arange(n, dtype=np.float64) / 100.0).reshape(-1, 1)
        return np.concatenate([states, states ** 2, al, al ** 2, al ** 3, np.ones((n, 1))], axis=1)

    def fit(self, states, returns):
        f = self._features(_np(states))
        a = f.T @ f + self.reg * np.eye(f.shape[1])
        self.weight = np.linalg.lstsq(a, f.T @ _np(returns), rcond=None)[0]

    def __call__(self, states):
        return self._features(_np(states)) @ self.weight


def compute_advantages(baseline, tau, gamma, rewards, dones, states, next_states, update_vf=True):
    """reference rl.py:95-110 (GAE with cherry semantics).  numpy float64 [N,1]."""
    rewards, dones = _np(rewards), _np(dones)
    returns = discount(gamma, rewards, dones)
    if update_vf:
        baseline.fit(states, returns)
    values, next_values = baseline(states), baseline(next_states)
    bootstraps = values * (1.0 - dones) + next_values * dones
    nxt = np.concatenate([bootstraps[1:], np.zeros((1, 1))], axis=0)
    td = rewards + gamma * (1.0 - dones) * nxt - bootstraps
    return discount(gamma * tau, td, dones)


def normalize(x, epsilon=1e-8):
    """cherry.normalize"""
    return (x - x.mean()) / (x.std(ddof=1) + epsilon) if x.size > 1 else x


def _advantages(ep, baseline, gamma, tau, update_vf=True):
    adv = compute_advantages(baseline, tau, gamma, ep['rewards'], ep['dones'], ep['states'], ep['next_states'], update_vf)
    return normalize(adv)


def _host_replays(replay_list):
    """Replays (dicts of device tensors) -> dicts of float64 numpy arrays with ONE device-to-host copy per field for the whole list
    (a copy per replay and field costs a stream synchronisation each: 240 of them for 20 tasks x 2 replays)."""
    keys = ('states', 'actions', 'rewards', 'dones', 'next_states')
    lens = [int(r['states'].shape[0]) for r in replay_list]
    out = [dict() for _ in replay_list]
    for k in keys:
        parts = [r[k].detach().reshape(r[k].shape[0], -1) if torch.is_tensor(r[k]) else torch.as_tensor(np.asarray(r[k])).reshape(len(r[k]), -1)
                 for r in replay_list]
        big = torch.cat([q.to(parts[0].device, torch.float64) for q in parts]).cpu().numpy()
        off = 0
        for o, n in zip(out, lens):
            o[k] = big[off:off + n]
            off += n
    return out


def _device_batch(replay_list, S, A, dev):
    """Replays (dicts of tensors / arrays) -> one padded fp32 device batch {states [R,B,S], actions [R,B,A], next_states, rewards
    [R,B], dones [R,B], count [R] int32}, assembled ON the device: one stack per field, no host round trip."""
    lens = [int(r['states'].shape[0]) for r in replay_list]
    B = max(lens)

    # fast path: every field of every replay already a contiguous fp32 tensor on the device -- mi_copy_segments packs all five fields of all
    # replays (padding rows zeroed) in one launch per 128 arrays, the host collects 5 R addresses (the concatenate / pad / gather per field
    # of the general path below is 0.5 ms of host time for the 40 ragged replays of a 20-task meta-iteration)
    packed = _device_batch_packed(replay_list, lens, B, S, A, dev)
    if packed is not None:
        return packed

    ragged = any(n != B for n in lens)
    if ragged:          # one row index for all fields: replay r's rows, then the appended zero row for its padding
        total = sum(lens)
        idx = np.full((len(lens), B), total, dtype=np.int64)
        off = 0
        for i, n in enumerate(lens):
            idx[i, :n] = np.arange(off, off + n)
            off += n
        idx = torch.from_numpy(idx.reshape(-1)).to(dev)
    first = replay_list[0]['states']
    plain = (not ragged and torch.is_tensor(first) and first.device == torch.device(dev) and
             all(torch.is_tensor(r[k]) and r[k].dtype == torch.float32 and r[k].device == first.device and r[k].is_contiguous()
                 for r in replay_list for k in ('states', 'actions', 'next_states', 'rewards', 'dones')))
    if plain:
        R = len(lens)
        out = {k: torch.stack([r[k].detach() for r in replay_list]).reshape(R, B, w)
               for k, w in (('states', S), ('actions', A), ('next_states', S))}
        out['rewards'] = torch.stack([r['rewards'].detach() for r in replay_list]).reshape(R, B)
        out['dones'] = torch.stack([r['dones'].detach() for r in replay_list]).reshape(R, B)
        out['count'] = torch.full((R,), B, dtype=torch.int32, device=dev)
        return out

    def field(k, width):
        parts = []
        for r, n in zip(replay_list, lens):
            t = r[k] if torch.is_tensor(r[k]) else torch.as_tensor(np.asarray(r[k]))
            parts.append(t.detach().to(dev, torch.float32).reshape(n, width))
        if not ragged:
            return torch.stack(parts).contiguous()
        # ragged replays (episodes that end early): concatenate, append a zero row, gather -- 3 launches per field instead of one
        # pad per replay
        flat = torch.nn.functional.pad(torch.cat(parts), (0, 0, 0, 1))
        return flat.index_select(0, idx).view(len(lens), B, width)

    out = dict(states=field('states', S), actions=field('actions', A), next_states=field('next_states', S),
               rewards=field('rewards', 1).reshape(len(lens), B), dones=field('dones', 1).reshape(len(lens), B))
    out['count'] = torch.tensor(lens, dtype=torch.int32, device=dev)
    return out


_FIELDS = ('states', 'actions', 'next_states', 'rewards', 'dones')


class Replay(dict):
    """A replay as the runners of this package return it: a dict of tensors that remembers the device addresses of its five fields once
    ``_device_batch_packed`` has checked them (fp32, contiguous, on the device), so the meta-step that gathers the same replays again
    reads one tuple per replay instead of inspecting five tensors.  Any change of the dict's entries forgets it; plain dicts work as before."""
    __slots__ = ('_mi_pack',)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._mi_pack = None

    def _forget(self):
        self._mi_pack = None

    def __setitem__(self, k, v):
        self._mi_pack = None
        super().__setitem__(k, v)

    def __delitem__(self, k):
        self._mi_pack = None
        super().__delitem__(k)

    def update(self, *a, **k):
        self._mi_pack = None
        super().update(*a, **k)

    def pop(self, *a):
        self._mi_pack = None
        return super().pop(*a)

    def popitem(self):
        self._mi_pack = None
        return super().popitem()

    def clear(self):
        self._mi_pack = None
        super().clear()

    def setdefault(self, *a):
        self._mi_pack = None
        return super().setdefault(*a)

    def __ior__(self, other):
        self._mi_pack = None
        return super().__ior__(other)

    def __reduce_ex__(self, protocol):
        # pickling (torch.save, multiprocessing) must not carry the remembered device ADDRESSES: the restored tensors live elsewhere
        return (Replay, (dict(self),))


def _device_batch_packed(replay_list, lens, B, S, A, dev):
    """_device_batch through mi_copy_segments, or None when a field is not a contiguous fp32 tensor on ``dev`` (the general path converts)."""
    dev = torch.device(dev)
    if dev.type != 'cuda':
        return None
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    R = len(lens)
    widths = (S, A, S, 1, 1)
    f32, Tensor, didx = torch.float32, torch.Tensor, dev.index
    srcs, cnts = [], []
    key = (didx, S, A)
    for r, n in zip(replay_list, lens):           # (attribute reads only: this loop is the host time of the call, ~0.8 us per array)
        memo = r._mi_pack if type(r) is Replay else None
        if memo is not None and memo[0] == key and memo[1] == n:
            srcs.extend(memo[2])                  # (checked before; the tensors are held by the replay, so the addresses are theirs)
            cnts.extend(memo[3])
            continue
        ps, cs = [], []
        for k, w in zip(_FIELDS, widths):
            t = r[k]
            if not (isinstance(t, Tensor) and t.dtype is f32 and t.is_cuda and t.get_device() == didx and t.is_contiguous() and t.numel() == n * w):
                return None
            ps.append(t.data_ptr())
            cs.append(n * w)
        if type(r) is Replay:
            r._mi_pack = (key, n, tuple(ps), tuple(cs))
        srcs.extend(ps)
        cnts.extend(cs)
    out = dict(states=torch.empty(R, B, S, dtype=f32, device=dev), actions=torch.empty(R, B, A, dtype=f32, device=dev),
               next_states=torch.empty(R, B, S, dtype=f32, device=dev), rewards=torch.empty(R, B, dtype=f32, device=dev),
               dones=torch.empty(R, B, dtype=f32, device=dev))
    base = [out[k].data_ptr() for k in _FIELDS]
    dsts, pads = [], []
    for i in range(R):
        for b, w in zip(base, widths):
            dsts.append(b + 4 * i * B * w)
            pads.append(B * w)
    from ..engine import copy_segments, upload_int32
    copy_segments(srcs, dsts, cnts, pads, dev)
    if all(n == B for n in lens):
        out['count'] = torch.full((R,), B, dtype=torch.int32, device=dev)
    else:
        out['count'] = upload_int32(lens, dev)
    return out


def _stacked_flat_parameters(policies, dev):
    """[len(policies), P] fp32: every policy's parameters in the engine's order (sigma first), gathered by mi_copy_segments from the
    parameter tensors where they lie; None when one of them is not a contiguous fp32 tensor on ``dev``."""
    dev = torch.device(dev)
    if dev.type != 'cuda':
        return None
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    f32, didx = torch.float32, dev.index
    srcs, cnts, offs, P = [], [], [], None
    for row, p in enumerate(policies):
        off = 0
        for q in p._engine_params():
            if not (q.dtype is f32 and q.is_cuda and q.get_device() == didx and q.is_contiguous()):
                return None
            n = q.numel()
            srcs.append(q.data_ptr())
            cnts.append(n)
            offs.append((row, off))                        # (the row travels with the segment: policies may split P differently)
            off += n
        if P is None:
            P = off
        elif off != P:
            return None
    out = torch.empty(len(policies), P, dtype=f32, device=dev)
    base = out.data_ptr()
    dsts = [base + 4 * (row * P + o) for row, o in offs]
    from ..engine import copy_segments
    copy_segments(srcs, dsts, cnts, cnts, dev)
    return out


def _advantages_device(batch, baseline, gamma, tau):
    """compute_advantages + ch.normalize (rl.py:95-110,355) of every replay of a device batch in one launch; the baseline is left
    fitted to the LAST replay, as after the reference's replay-by-replay walk."""
    from ..engine import gae_advantages
    adv, wts = gae_advantages(batch['states'], batch['next_states'], batch['rewards'], batch['dones'], batch['count'], gamma, tau,
                              baseline.reg, normalize=True, want_weights=True)
    baseline._weight_dev = wts[-1]
    return adv


def _replay_on_device(ep, baseline, gamma, tau, S, A, dev, normalize=True, update_vf=True):
    """One replay as a device batch with its advantages: {states [1,B,S], actions, adv [1,B], count, done} (mi_gae_advantages;
    update_vf=False evaluates the baseline as last fitted, rl.py:401)."""
    from ..engine import gae_advantages
    b = _device_batch([ep], S, A, dev)
    weights = None
    if not update_vf:
        weights = baseline._weight_dev if baseline._weight_dev is not None else torch.from_numpy(np.ascontiguousarray(baseline.weight)).reshape(1, -1)
    adv, wts = gae_advantages(b['states'], b['next_states'], b['rewards'], b['dones'], b['count'], gamma, tau, baseline.reg,
                              normalize=normalize, want_weights=True, weights=weights)
    if update_vf:
        baseline._weight_dev = wts[-1]
    return dict(states=b['states'], actions=b['actions'], adv=adv, count=b['count'], done=b['dones'])


def _gae_on_device(dev, S, rows):
    from ..engine import gae_max_rows
    return torch.device(dev).type == 'cuda' and 0 < rows <= gae_max_rows(S)


def _pad(eps_list, advs, S, A, dev):
    """List of replays (one per task) -> padded device batch {states [T,B,S], actions, adv, count}: assembled on the host, one
    upload per field."""
    T = len(eps_list)
    B = max(int(e['states'].shape[0]) for e in eps_list)
    if advs and torch.is_tensor(advs[0]) and advs[0].is_cuda:      # advantages already on the device (mi_gae_advantages): stay there
        b = _device_batch(eps_list, S, A, dev)
        adv = torch.stack([torch.nn.functional.pad(a.reshape(-1).float(), (0, B - a.numel())) for a in advs]).contiguous()
        return dict(states=b['states'], actions=b['actions'], adv=adv, count=b['count'], done=b['dones'])
    st, ac = np.zeros((T, B, S), np.float32), np.zeros((T, B, A), np.float32)
    ad, cnt, dn = np.zeros((T, B), np.float32), np.zeros(T, np.int32), np.zeros((T, B), np.float32)
    for t, (e, a) in enumerate(zip(eps_list, advs)):
        n = int(e['states'].shape[0])
        st[t, :n] = _np(e['states']).reshape(n, S)
        ac[t, :n] = _np(e['actions']).reshape(n, A)
        ad[t, :n] = np.asarray(a, dtype=np.float32).reshape(-1)
        if 'dones' in e:
            dn[t, :n] = _np(e['dones']).reshape(n)
        cnt[t] = n
    up = lambda x: torch.from_numpy(x).to(dev)
    return dict(states=up(st), actions=up(ac), adv=up(ad), count=up(cnt), done=up(dn))


# ---------------------------------------------------------------------------------------------- reference functions
def trpo_a2c_loss(episodes, learner, baseline, gamma, tau, update_vf=True):
    """reference rl.py:346-358: -mean(log_prob * normalised advantages) (value only; the gradient path is trpo_update)."""
    episodes = _as_replay(episodes)
    dev = learner.sigma.device
    if _gae_on_device(dev, learner.input_size, int(episodes['states'].shape[0])):
        b = _replay_on_device(episodes, baseline, gamma, tau, learner.input_size, learner.output_size, dev, update_vf=update_vf)
        lp = learner.log_prob(b['states'][0], b['actions'][0])
        return -(lp * b['adv'][0].reshape(-1, 1).to(lp)).mean()
    adv = _advantages(episodes, baseline, gamma, tau, update_vf)
    lp = learner.log_prob(episodes['states'].to(dev), episodes['actions'].to(dev))
    return -(lp * torch.from_numpy(adv).to(lp)).mean()


def trpo_update(episodes, learner, baseline, inner_lr, gamma, tau, anil=False, first_order=False):
    """reference rl.py:361-374: one MAML update of the policy on ``episodes``; returns the adapted policy (a new object; the
    second-order dependence on the original parameters is handled inside meta_optimize_trpo's fused calls)."""
    # anil only sets allow_unused (rl.py:371): which parameters move is decided by the policy's own switch -- with
    # DiagNormalPolicyANIL.turn_off_body_grads() the body's gradients are None and maml_update leaves it unchanged.
    episodes = _as_replay(episodes)
    head_only = bool(getattr(learner, 'features_no_grad', False))
    eng = learner.engine()
    dev = learner.sigma.device
    if _gae_on_device(dev, learner.input_size, int(episodes['states'].shape[0])):
        batch = _replay_on_device(episodes, baseline, gamma, tau, learner.input_size, learner.output_size, dev)
    else:
        batch = _pad([episodes], [_advantages(episodes, baseline, gamma, tau)], learner.input_size, learner.output_size, dev)
    theta_new, _ = eng.adapt(learner.flat(), batch['states'], batch['actions'], batch['adv'], batch['count'], inner_lr,
                             head_only=head_only)
    new = deepcopy(learner)
    new.load_flat(theta_new[0])
    return new


def fast_adapt_trpo(task, learner, baseline, params, anil=False, first_order=False, render=False):
    """reference rl.py:377-406.  ``task.run(policy, episodes=n)`` returns a replay (dict or cherry-style object, ``_as_replay``).
    ``learner``: the policy itself or a ``MAML`` wrapper around it (rl/anil_trpo.py:84 wraps; rl.py:382,396 reach the policy as
    ``learner.module``)."""
    task_replay = []
    if anil:                                                   # rl.py:381-382
        _unwrap(learner).turn_off_body_grads()
    for step in range(params['adapt_steps']):
        support_episodes = task.run(learner, episodes=params['adapt_batch_size'])
        task_replay.append(support_episodes)
        learner = trpo_update(support_episodes, learner, baseline, params['inner_lr'], params['gamma'], params['tau'],
                              anil=anil, first_order=first_order)
    if anil:                                                   # rl.py:395-396
        _unwrap(learner).turn_on_body_grads()
    query_episodes = task.run(learner, episodes=params['adapt_batch_size'])
    task_replay.append(query_episodes)
    valid_loss = trpo_a2c_loss(query_episodes, learner, baseline, params['gamma'], params['tau'], update_vf=False)
    query_rew = _as_replay(query_episodes)['rewards'].sum().item() / params['adapt_batch_size']                    # rl.py:403
    query_success_rate = get_ep_successes(query_episodes, params.get('max_path_length')) / params['adapt_batch_size']  # rl.py:404
    return learner, valid_loss, task_replay, query_rew, query_success_rate


class _SurrogateContext:
    """Everything meta_surrogate_loss needs that does not depend on the candidate parameters (advantages are functions of
    the replays only -- the reference re-fits the baseline to the same data on every call, rl.py:99,465)."""

    def __init__(self, iter_replays, iter_policies, policy, baseline, params):
        S, A, dev = policy.input_size, policy.output_size, policy.sigma.device
        iter_replays = [[_as_replay(r) for r in task] for task in iter_replays]
        K = len(iter_replays[0]) - 1
        if K < 1 or any(len(r) != K + 1 for r in iter_replays):
            raise ValueError('every task needs the same number (>= 1) of support replays plus one query replay')
        self.steps = K
        adv = lambda e: _advantages(e, baseline, params['gamma'], params['tau'])
        nT = len(iter_replays)
        flat = [r[k] for k in range(K + 1) for r in iter_replays]                    # [k][task]
        if _gae_on_device(dev, S, max(int(r['states'].shape[0]) for r in flat)):
            # advantages of all (K + 1) x tasks replays in ONE launch, the batch assembled on the device (the reference re-fits the
            # baseline and re-runs GAE per task and replay on the CPU at every evaluation of the surrogate, rl.py:444-465)
            batch = _device_batch(flat, S, A, dev)
            advs = _advantages_device(batch, baseline, params['gamma'], params['tau'])
            part = lambda k: dict(states=batch['states'][k * nT:(k + 1) * nT], actions=batch['actions'][k * nT:(k + 1) * nT],
                                  adv=advs[k * nT:(k + 1) * nT], count=batch['count'][k * nT:(k + 1) * nT],
                                  done=batch['dones'][k * nT:(k + 1) * nT])
            sups = [part(k) for k in range(K)]
            self.qry = part(K)
        else:
            host = _host_replays(flat)                               # one D2H copy per field
            sups = []
            for k in range(K):                                       # the reference walks task by task, replay by replay;
                eps = host[k * nT:(k + 1) * nT]                      # the baseline is re-fitted per replay, so the order is immaterial
                sups.append(_pad(eps, [adv(e) for e in eps], S, A, dev))
            qry_eps = host[K * nT:]
            self.qry = _pad(qry_eps, [adv(e) for e in qry_eps], S, A, dev)
        B = max([d['states'].shape[1] for d in sups] + [self.qry['states'].shape[1]])
        for d in sups + [self.qry]:                              # one common padded length
            if d['states'].shape[1] < B:
                pad = B - d['states'].shape[1]
                d['states'] = torch.nn.functional.pad(d['states'], (0, 0, 0, pad))
                d['actions'] = torch.nn.functional.pad(d['actions'], (0, 0, 0, pad))
                d['adv'] = torch.nn.functional.pad(d['adv'], (0, pad))
            for k in ('states', 'actions', 'adv'):
                d[k] = d[k].contiguous()
        if K == 1:
            self.sup = sups[0]
        else:
            self.sup = {k: torch.stack([d[k] for d in sups]).contiguous() for k in ('states', 'actions', 'adv', 'count')}
        self.engine = policy.engine()
        # the stored old policies' parameters as ONE gather ([tasks, P], engine order: sigma first) and their scales from its first columns
        # (per-policy flat() / clamp / exp launches were ~80 of the ~100 launches of this constructor)
        if all(hasattr(p, '_engine_params') for p in iter_policies):
            thetas = _stacked_flat_parameters(iter_policies, dev)
            if thetas is None:
                thetas = torch.cat([q.detach().reshape(-1).float() for p in iter_policies for q in p._engine_params()]).view(len(iter_policies), -1)
        else:                                                       # any object with the policy protocol (flat(), sigma first)
            thetas = torch.stack([p.flat() for p in iter_policies])
        self.old_loc = self.engine.forward(thetas, self.qry['states'])
        self.old_scale = torch.exp(torch.clamp(thetas[:, :A], min=float(np.log(1e-6)))).contiguous()
        self.inner_lr = params['inner_lr']

    # Multi-GPU (TRPO): every rank holds a shard of the task list; the mean over tasks of (loss, kl, grad) and
    # of each Fisher-vector product is completed by one small all-reduce per evaluation (RCCL; 42 KB for the 2x100 policy).
    def _allmean(self, *tensors):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return tensors
        n_local = float(self.qry['states'].shape[0])
        flat = torch.cat([t.reshape(-1) for t in tensors] + [torch.ones(1, device=tensors[0].device)]) * n_local
        dist.all_reduce(flat)
        flat = flat / flat[-1]
        out, off = [], 0
        for t in tensors:
            out.append(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        return tuple(out)

    def evaluate(self, theta, want_grad=False):
        fn = self.engine.surrogate if self.steps == 1 else self.engine.surrogate_steps
        loss, kl, grad = fn(theta, self.sup, self.qry, self.old_loc, self.old_scale, self.inner_lr, want_grad)
        if grad is None:
            loss, kl = self._allmean(loss, kl)
        else:
            loss, kl, grad = self._allmean(loss, kl, grad)
        return loss, kl, grad

    def prepare_general_kl(self, theta):
        """ANIL-TRPO: the re-adapted policies differ from the stored old ones, the Fisher form does not apply; set up the exact
        KL Hessian-vector product (mi_trpo_kl_prepare) at the theta of the preceding ``evaluate``."""
        if self.steps != 1:
            raise NotImplementedError('the exact KL Hessian-vector product (anil=True) is implemented for adapt_steps == 1, the '
                                      'reference default (rl/anil_trpo.py)')
        self.engine.kl_prepare(theta, self.sup, self.qry, self.old_loc, self.old_scale, self.inner_lr)
        self.general = True

    def fvp(self, theta, v, damping=1e-5):
        # the damping term is linear in v, so averaging the per-rank results keeps it exact
        if getattr(self, 'general', False):
            return self._allmean(self.engine.fvp_general(theta, self.sup, self.qry, self.old_scale, self.inner_lr, damping, v))[0]
        if self.steps == 1:
            return self._allmean(self.engine.fvp(theta, self.sup, self.qry, self.inner_lr, damping, v))[0]
        return self._allmean(self.engine.fvp_steps(self.sup, self.qry, self.inner_lr, damping, v))[0]


def meta_surrogate_loss(iter_replays, iter_policies, policy, baseline, params, anil=False):
    """reference rl.py:441-473 -> (mean surrogate loss, mean KL) as 0-dim tensors."""
    ctx = _SurrogateContext(iter_replays, iter_policies, policy, baseline, params)
    loss, kl, _ = ctx.evaluate(policy.flat())
    return loss[0], kl[0]


def conjugate_gradient(Ax, b, num_iterations=10, tol=1e-10, eps=1e-8):
    """cherry.algorithms.trpo.conjugate_gradient (reference rl.py:418).  The recurrences (dot products, axpys on 10,604
    elements) are carried in fp64 on the device; every A p is one fp32 Fisher-vector product through the engine."""
    dt = b.dtype
    if b.is_cuda:
        return _conjugate_gradient_device(Ax, b, num_iterations, tol, eps)
    b = b.double()
    x = torch.zeros_like(b)
    r, p = b.clone(), b.clone()
    r_dot_old = torch.dot(r, r)
    for _ in range(num_iterations):
        Ap = Ax(p.to(dt)).double()
        alpha = r_dot_old / (torch.dot(p, Ap) + eps)
        x += alpha * p
        r -= alpha * Ap
        r_dot_new = torch.dot(r, r)
        p = r + (r_dot_new / r_dot_old) * p
        r_dot_old = r_dot_new
        if r_dot_new.item() < tol:
            break
    return x.to(dt)


def _conjugate_gradient_device(Ax, b, num_iterations, tol, eps):
    """The same recurrences with the whole loop body after `Ap = Ax(p)` as one launch (mi_cg_update_checked)."""
    from .. import _lib
    from ..engine import _ptr, _stream
    lib = _lib.load()
    dev, n = b.device, b.numel()
    if b.dtype == torch.float32:
        b32 = b.detach().reshape(-1).contiguous()
        x, r, p = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
        p32 = torch.empty(n, dtype=torch.float32, device=dev)
        rr = torch.empty(3, dtype=torch.float64, device=dev)      # r.r, last step length, "converged" latch
        with torch.cuda.device(dev):                              # x = 0, r = p = b, rr = (b.b, 0, 0): one launch
            _lib.check(lib.mi_cg_init(_stream(dev), _ptr(b32), _ptr(x), _ptr(r), _ptr(p), _ptr(p32), _ptr(rr), n))
    else:
        r = b.detach().to(torch.float64, copy=True).reshape(-1).contiguous()      # a copy: mi_cg_update overwrites r in place
        x, p = torch.zeros_like(r), r.clone()
        p32 = r.float()
        rr = torch.zeros(3, dtype=torch.float64, device=dev)
        rr[0] = torch.dot(r, r)
    for _ in range(num_iterations):
        # the reference's `if r_dot_new < tol: break` is taken on the device (mi_cg_update_checked): after the break every later
        # recurrence is a no-op, so the loop needs no host synchronisation per iteration and x is exactly the x at the break
        ap = Ax(p32.to(b.dtype)).detach().float().reshape(-1).contiguous()
        with torch.cuda.device(dev):
            _lib.check(lib.mi_cg_update_checked(_stream(dev), _ptr(x), _ptr(r), _ptr(p), _ptr(ap), _ptr(rr), _ptr(p32), n, float(eps), float(tol)))
    return x.to(b.dtype).reshape(b.shape)


def meta_optimize_trpo(params, policy, baseline, iter_replays, iter_policies, anil=False):
    """reference rl.py:409-438: CG step direction from the Fisher-vector product of the mean KL, then backtracking line
    search on (surrogate loss, KL); updates ``policy`` in place.  Returns diagnostics."""
    ctx = _SurrogateContext(iter_replays, iter_policies, policy, baseline, params)
    theta = policy.flat()
    old_loss, old_kl, grad = ctx.evaluate(theta, want_grad=True)
    if anil:
        # The reference's surrogate replays the inner step with ALL parameters (clone_module(policy) has its body grads on,
        # rl.py:447-453) while the stored old policies were adapted head-only (rl.py:381-382): at the current parameters the new
        # policies differ from the old ones, KL's gradient is not zero, and trpo.hessian_vector_product(kl) is the exact Hessian
        # of the mean KL including the third derivative of the inner loss (mi_trpo_fvp_general).
        ctx.prepare_general_kl(theta)
    Fvp = lambda v: ctx.fvp(theta, v)
    step = conjugate_gradient(Fvp, grad)
    if step.is_cuda and step.dtype == torch.float32:
        # shs = 0.5 step . F step, lagrange = sqrt(shs / max_kl), step / lagrange (rl.py:419-421) as one launch
        from .. import _lib
        from ..engine import _ptr, _stream
        step, fstep = step.contiguous(), Fvp(step).detach().float().reshape(-1).contiguous()
        scaled = torch.empty_like(step)
        with torch.cuda.device(step.device):
            _lib.check(_lib.load().mi_trpo_scale_step(_stream(step.device), _ptr(step), _ptr(fstep), step.numel(), float(params['max_kl']),
                                                      _ptr(scaled), None))
        step = scaled
    else:
        shs = 0.5 * torch.dot(step, Fvp(step))
        lagrange_multiplier = torch.sqrt(shs / params['max_kl'])
        step = step / lagrange_multiplier
    accepted, new_loss, kl = None, None, None
    for ls_step in range(params['ls_max_steps']):
        stepsize = params['backtrack_factor'] ** ls_step * params['outer_lr']
        cand = torch.add(theta, step, alpha=-stepsize).contiguous()
        new_loss, kl, _ = ctx.evaluate(cand)
        # (one host read for the three scalars of the test: each .item() is a copy and a synchronisation of its own, ~25 us)
        nl, ol, klv = torch.cat([new_loss.reshape(-1)[:1], old_loss.reshape(-1)[:1], kl.reshape(-1)[:1]]).tolist()
        if nl < ol and klv < params['max_kl']:
            policy.load_flat(cand)
            accepted = ls_step
            break
    return dict(grad=grad, step=step, old_loss=old_loss[0], old_kl=old_kl[0], accepted=accepted,
                new_loss=None if new_loss is None else new_loss[0], kl=None if kl is None else kl[0], fvp=Fvp, context=ctx)


# ---------------------------------------------------------------------------------------------- VPG / PPO (reference rl.py:209-337)
def _unwrap(learner):
    return getattr(learner, 'module', learner)


class _PolicyMetaLoss(torch.autograd.Function):
    """Validation loss of one task whose gradient w.r.t. the policy parameters came out of the same fused call
    (mi_policy_meta_batch); `.backward()` accumulates it like the reference's autograd graph through learner.adapt."""

    @staticmethod
    def forward(ctx, loss, grad, shapes, *params):
        ctx.shapes = shapes
        ctx.save_for_backward(grad)
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        outs, off = [], 0
        for shp in ctx.shapes:
            n = int(torch.Size(shp).numel())
            outs.append((grad[off:off + n] * gout).reshape(shp))
            off += n
        return (None, None, None) + tuple(outs)


def _stack(replays, advs, S, A, dev):
    """Support replays of ONE task -> {states [NB,1,B,S], ...} with a common padded length."""
    batches = [_pad([e], [a], S, A, dev) for e, a in zip(replays, advs)]
    B = max(b['states'].shape[1] for b in batches)
    def padto(t, dim):
        n = B - t.shape[dim]
        if n == 0:
            return t
        pad = [0, 0] * (t.dim() - 1 - dim) + [0, n]
        return torch.nn.functional.pad(t, pad)
    out = {k: torch.stack([padto(b[k], 1) if k != 'count' else b[k] for b in batches]).contiguous()
           for k in ('states', 'actions', 'adv', 'count', 'done')}
    return out, B


def _replay_meta(pol, support, sup_adv, query, q_adv, inner_lr, loss, clip, epochs, anil, first_order, with_grad):
    """One task through mi_policy_meta_batch: (validation loss [1], adapted theta [P], grad [P] or None)."""
    S, A, dev = pol.input_size, pol.output_size, pol.sigma.device
    sup, B1 = _stack(support, sup_adv, S, A, dev) if support else (None, 0)
    qry = _pad([query], [q_adv], S, A, dev)
    B = max(B1, qry['states'].shape[1])
    def fit(d, dim):
        for k in ('states', 'actions', 'adv', 'done'):
            n = B - d[k].shape[dim]
            if n:
                pad = [0, 0] * (d[k].dim() - 1 - dim) + [0, n]
                d[k] = torch.nn.functional.pad(d[k], pad).contiguous()
        return d
    if sup is not None:
        sup = fit(sup, 2)
    qry = fit(qry, 1)
    step_batch = [b for b in range(len(support)) for _ in range(epochs)]
    lt, th, g = pol.engine().meta_batch(pol.flat(), sup, qry, step_batch, inner_lr, loss=loss, clip=clip, head_only=anil,
                                        first_order=first_order, with_grad=with_grad)
    return lt, th[0], g


def _adapt_and_validate(task, learner, baseline, params, algo, anil, first_order, dice=False):
    pol = _unwrap(learner)
    gamma, tau = params['gamma'], params['tau']
    epochs = params['ppo_epochs'] if algo == 'ppo' else 1
    clip = params.get('ppo_clip_ratio', 0.1)
    kind = 'ppo' if algo == 'ppo' else ('dice' if dice else 'a2c')

    def adv_of(ep):
        n = int(ep['states'].shape[0])
        if _gae_on_device(pol.sigma.device, pol.input_size, n):         # stays on the device: [n] fp32
            return _replay_on_device(ep, baseline, gamma, tau, pol.input_size, pol.output_size, pol.sigma.device,
                                     normalize=(algo == 'ppo'))['adv'][0, :n]
        a = compute_advantages(baseline, tau, gamma, ep['rewards'], ep['dones'], ep['states'], ep['next_states'])
        return normalize(a) if algo == 'ppo' else a                     # vpg_a2c_loss does not normalise (rl.py:217)

    if anil:
        pol.turn_off_body_grads()
    support, sup_adv, current = [], [], pol
    for step in range(params['adapt_steps']):
        ep = _as_replay(task.run(current, episodes=params['adapt_batch_size']))
        support.append(ep)
        sup_adv.append(adv_of(ep))
        # parameters after the updates so far (the rollouts of the next step / the query need them); value of the loss unused
        _, theta_k, _ = _replay_meta(pol, support, sup_adv, ep, sup_adv[-1], params['inner_lr'], kind, clip, epochs, anil, first_order, False)
        current = deepcopy(pol)
        current.load_flat(theta_k)
    if anil:
        pol.turn_on_body_grads()
    query = _as_replay(task.run(current, episodes=params['adapt_batch_size']))
    need = torch.is_grad_enabled() and any(q.requires_grad for q in pol.parameters())
    lt, _, grad = _replay_meta(pol, support, sup_adv, query, adv_of(query), params['inner_lr'], kind, clip, epochs, anil, first_order, need)
    if need:
        eparams = pol._engine_params()
        valid_loss = _PolicyMetaLoss.apply(lt[0], grad, [q.shape for q in eparams], *eparams)
    else:
        valid_loss = lt[0]
    rew = query['rewards'].sum().item() / params['adapt_batch_size']
    suc = get_ep_successes(query, params.get('max_path_length')) / params['adapt_batch_size']                          # rl.py:252,315
    # learn2learn's learner.adapt updates the learner in place; here the base module is shared by every clone, so the adapted
    # parameters are handed over on the side (evaluate() below acts with them)
    try:
        learner._adapted_policy = current
    except Exception:
        pass
    return valid_loss, rew, suc


def fast_adapt_vpg(task, learner, baseline, params, anil=False, first_order=False, render=False, dice=False):
    """reference rl.py:231-255 -> (valid_loss, query reward, success rate); `valid_loss.backward()` accumulates the MAML
    gradient (second order unless first_order) into the policy's parameters.  ``dice`` (not a parameter of the reference's
    fast_adapt_vpg, whose calls leave vpg_a2c_loss at dice=False): every vpg_a2c_loss of the adaptation -- the adapt losses and
    the validation loss -- uses the DiCE objective of rl.py:219-226."""
    return _adapt_and_validate(task, learner, baseline, params, 'vpg', anil, first_order, dice=dice)


def vpg_a2c_loss(episodes, learner, baseline, gamma, tau, dice=False):
    """reference rl.py:208-228, value only (the gradient path is the fused fast_adapt_vpg): -mean(log_prob * advantages), or with
    ``dice`` -mean(magic_box(weighted_cumsum(log_probs, weights)) * advantages) = -mean(advantages) (magic_box evaluates to 1)."""
    episodes = _as_replay(episodes)
    adv = compute_advantages(baseline, tau, gamma, episodes['rewards'], episodes['dones'], episodes['states'], episodes['next_states'])
    adv_t = torch.from_numpy(adv).to(device=device, dtype=torch.float32)
    if dice:
        return -adv_t.mean()
    lp = _unwrap(learner).log_prob(episodes['states'].to(device), episodes['actions'].to(device))
    return -(lp * adv_t).mean()


def fast_adapt_ppo(task, learner, baseline, params, anil=False, render=False):
    """reference rl.py:267-318: ppo_epochs clipped-surrogate updates per adapt step (second order, as learner.adapt defaults)."""
    return _adapt_and_validate(task, learner, baseline, params, 'ppo', anil, False)


def _eval_tasks(env, params, goals):
    """The evaluation tasks and a runner factory for them.  ``env`` as the reference passes it (rl.py:142-196: an environment NAME handed to
    make_env, whose result offers ``sample_tasks / set_task / reset``): 'Particles2D-v1' (or any name containing 'Particles2D') builds this
    package's ``Particles2DEnv``; an env-like object is used as it is -- ``Particles2DEnv`` through the device-vectorised
    ``Particles2DRunner``, anything else through the host loop of ``EnvRunner``; a sequence of goals (this package's earlier signature, also
    the ``goals=`` keyword) is taken as the task list itself."""
    if goals is None and not isinstance(env, str) and not hasattr(env, 'sample_tasks'):
        goals, env = env, None
    if isinstance(env, str):
        if 'Particles2D' not in env:
            raise NotImplementedError(f'environment {env!r}: this package ships Particles2D (Meta-World / MuJoCo are out of scope); pass an '
                                      'env-like object with sample_tasks / set_task / reset / step instead of a name')
        env = Particles2DEnv(seed=params.get('seed', 42))
    if goals is not None:
        tasks = [{'goal': g} for g in goals]
        env = env if env is not None else Particles2DEnv(seed=params.get('seed', 42))
    else:
        tasks = env.sample_tasks(params['n_tasks'])                                    # rl.py:162
    return env, tasks


def evaluate(algo, env, policy, baseline, params, anil=False, render=False, generator=None, goals=None):
    """reference rl.py:142-196: adapt a copy of the policy to every evaluation task with `algo` in {'vpg', 'ppo', 'trpo'}, then roll out
    `adapt_batch_size` query episodes with the adapted policy.  ``env``: see ``_eval_tasks`` (name, env-like, or the task goals).
    Returns (tasks_rewards, mean reward, mean success rate)."""
    env, tasks = _eval_tasks(env, params, goals)
    dev = _unwrap(policy).sigma.device
    tasks_rewards, tasks_success = [], []
    for task_desc in tasks:
        learner = deepcopy(policy)
        env.set_task(task_desc)                                                        # rl.py:166-167
        env.reset()
        if isinstance(env, Particles2DEnv):
            task = Particles2DRunner(env.goal, params['max_path_length'], generator, dev)
        else:
            task = EnvRunner(env, params['max_path_length'], dev)
        with torch.no_grad():
            if algo == 'vpg':
                fast_adapt_vpg(task, learner, baseline, params, anil=anil)
                adapted = learner._adapted_policy
            elif algo == 'ppo':
                fast_adapt_ppo(task, learner, baseline, params)
                adapted = learner._adapted_policy
            else:
                adapted, _, _, _, _ = fast_adapt_trpo(task, _unwrap(learner), baseline, params, anil=anil)
        query = _as_replay(task.run(adapted, episodes=params['adapt_batch_size']))    # rl.py:181-184
        tasks_rewards.append(query['rewards'].sum().item() / params['adapt_batch_size'])
        tasks_success.append(get_ep_successes(query, params.get('max_path_length')) / params['adapt_batch_size'])
    n = params.get('n_tasks', len(tasks_rewards))
    if isinstance(n, str):                                                             # (rl.py:158-159: an explicit task name)
        n = len(tasks_rewards)
    return tasks_rewards, sum(tasks_rewards) / n, sum(tasks_success) / n


def evaluate_vpg(env, policy, baseline, eval_params, anil=False, render=False, generator=None, goals=None):
    """reference rl.py:258-259"""
    return evaluate('vpg', env, policy, baseline, eval_params, anil, render, generator, goals)


def evaluate_ppo(env, policy, baseline, eval_params, anil=False, render=False, generator=None, goals=None):
    """reference rl.py:340-341"""
    return evaluate('ppo', env, policy, baseline, eval_params, anil, render, generator, goals)


def evaluate_trpo(env, policy, baseline, eval_params, anil=False, render=False, generator=None, goals=None):
    """reference rl.py:476-477"""
    return evaluate('trpo', env, policy, baseline, eval_params, anil, render, generator, goals)


# ---------------------------------------------------------------------------------------------- Particles2D rollouts (host loop, device math)
class Particles2DRunner:
    """Minimal stand-in for core_functions/runner.py + learn2learn's Particles2D (both out of scope):
    all ``episodes`` of a task advance in lock-step on the device; ``run`` returns a replay dict with episodes concatenated."""

    def __init__(self, goal, max_path_length, generator=None, dev=None):
        self.dev = dev or device
        self.goal = torch.as_tensor(goal, dtype=torch.float32, device=self.dev)
        self.max_path_length, self.generator = max_path_length, generator

    def run(self, policy, episodes):
        E, L = episodes, self.max_path_length
        state = torch.zeros(E, 2, device=self.dev)
        active = torch.ones(E, dtype=torch.bool, device=self.dev)
        S, A, R, D, NS, M = [], [], [], [], [], []
        scale = torch.exp(torch.clamp(policy.sigma.detach(), min=np.log(1e-6)))
        theta, eng = policy.flat(), policy.engine()
        for t in range(L):
            loc = eng.forward(theta, state.unsqueeze(0))[0]
            action = loc + scale * torch.randn(loc.shape, device=self.dev, generator=self.generator)
            nxt = state + action.clamp(-0.1, 0.1)
            diff = nxt - self.goal
            reward = -diff.norm(dim=1)
            done = (diff.abs() < 0.01).all(dim=1)
            last = done | (t == L - 1)
            S.append(state); A.append(action); R.append(reward); D.append(last.float()); NS.append(nxt); M.append(active)
            state = nxt
            active = active & ~done
        S, A, NS = torch.stack(S, 1), torch.stack(A, 1), torch.stack(NS, 1)          # [E, L, *]
        R, D, M = torch.stack(R, 1), torch.stack(D, 1), torch.stack(M, 1)
        keep = M.reshape(-1)
        flat = lambda x: x.reshape(E * L, -1)[keep]
        return Replay(states=flat(S), actions=flat(A), rewards=flat(R), dones=flat(D), next_states=flat(NS))


class Particles2DEnv:
    """learn2learn's Particles2D as the reference's drivers use it (rl/maml_trpo.py:100-108: ``sample_tasks / set_task / reset``; out of
    scope as a dependency): state in R^2 from the origin, goal ~ U(-0.5, 0.5)^2, action clipped to +-0.1,
    reward = -||state - goal||_2, done when both |state - goal| < 0.01.  Episodes are rolled out on the device by ``Particles2DRunner``
    (``runner()``); ``step`` is the same arithmetic one step at a time on the host, for callers that drive the env themselves."""
    state_size, action_size = 2, 2

    def __init__(self, seed=42):
        self.rng = np.random.RandomState(seed)
        self.goal = np.zeros(2, dtype=np.float32)
        self.state = np.zeros(2, dtype=np.float32)

    def sample_tasks(self, num_tasks):
        return [{'goal': g} for g in self.rng.uniform(-0.5, 0.5, size=(num_tasks, 2))]

    def set_task(self, task):
        self.goal = np.asarray(task['goal'], dtype=np.float32)

    def reset(self):
        self.state = np.zeros(2, dtype=np.float32)
        return self.state.copy()

    def step(self, action):
        self.state = self.state + np.clip(np.asarray(action, dtype=np.float32).reshape(2), -0.1, 0.1)
        diff = self.state - self.goal
        return self.state.copy(), -float(np.sqrt((diff * diff).sum())), bool((np.abs(diff) < 0.01).all()), {}

    def runner(self, max_path_length, generator=None, dev=None):
        return Particles2DRunner(self.goal, max_path_length, generator, dev)


class EnvRunner:
    """``task.run(policy, episodes=n)`` for ANY env-like object (``reset() -> state``, ``step(action) -> (state, reward, done, info)``):
    the host loop of the reference's core_functions/runner.py for one environment, the policy evaluated on the device one state at a
    time.  ``info['success']`` (Meta-World's signal, runner.py's ``extra_info``) is recorded as the replay's ``success`` when present.
    Slow by construction (an environment step per policy call); Particles2D has its own vectorised runner."""

    def __init__(self, env, max_path_length, dev=None):
        self.env, self.max_path_length, self.dev = env, max_path_length, dev or device

    def run(self, policy, episodes, render=False):
        S, A, R, D, NS, SU = [], [], [], [], [], []
        any_success = False
        for _ in range(episodes):
            state = np.asarray(self.env.reset(), dtype=np.float32).reshape(-1)
            for t in range(self.max_path_length):
                st = torch.from_numpy(state).to(self.dev).reshape(1, -1)
                action = policy(st)[0]
                nxt, reward, done, info = self.env.step(action.detach().cpu().numpy())
                nxt = np.asarray(nxt, dtype=np.float32).reshape(-1)
                last = bool(done) or t == self.max_path_length - 1
                S.append(st[0]); A.append(action.detach().float()); R.append(float(reward)); D.append(1.0 if last else 0.0)
                NS.append(torch.from_numpy(nxt).to(self.dev))
                if isinstance(info, dict) and 'success' in info:
                    any_success = True
                    SU.append(float(info['success']))
                else:
                    SU.append(0.0)
                state = nxt
                if done:
                    break
        col = lambda x: torch.tensor(x, dtype=torch.float32, device=self.dev).reshape(-1, 1)
        out = Replay(states=torch.stack(S).contiguous(), actions=torch.stack(A).contiguous(), rewards=col(R), dones=col(D),
                     next_states=torch.stack(NS).contiguous())
        if any_success:
            out['success'] = col(SU)
        return out
