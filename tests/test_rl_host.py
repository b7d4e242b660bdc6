"""Host-side RL pieces of the product (discount / LinearValue / GAE / normalize in core_functions/rl.py) against the oracle's
restatement of the same cherry semantics."""
import numpy as np
import torch

from exploring_meta_amd.core_functions import rl as PR
from oracle import rl_ref as RL


def _replay(seed, n_eps=5, length=17):
    g = torch.Generator().manual_seed(seed)
    n = n_eps * length
    dones = torch.zeros(n, 1, dtype=torch.float64)
    dones[length - 1::length] = 1.0
    dones[7] = 1.0                                   # one early termination
    return dict(states=torch.randn(n, 2, generator=g, dtype=torch.float64), actions=torch.randn(n, 2, generator=g, dtype=torch.float64),
                rewards=-torch.rand(n, 1, generator=g, dtype=torch.float64), dones=dones,
                next_states=torch.randn(n, 2, generator=g, dtype=torch.float64))


def test_advantages_match_oracle():
    ep = _replay(0)
    for update_vf in (True,):
        a = PR.compute_advantages(PR.LinearValue(2, 2), 1.0, 0.99, ep['rewards'], ep['dones'], ep['states'], ep['next_states'], update_vf)
        b = RL.compute_advantages(RL.LinearValue(2, 2), 1.0, 0.99, ep, update_vf)
        assert np.allclose(a, b.numpy(), rtol=1e-9, atol=1e-10)
        assert np.allclose(PR.normalize(a), RL.normalize(b).numpy(), rtol=1e-9, atol=1e-10)


def test_discount_matches_oracle():
    ep = _replay(1)
    assert np.allclose(PR.discount(0.9, ep['rewards'].numpy(), ep['dones'].numpy()), RL.discount(0.9, ep['rewards'], ep['dones']).numpy())


def test_replay_forgets_its_addresses_when_an_entry_changes():
    """core_functions.rl.Replay: the dict the runners return; every way of changing its entries drops the remembered device addresses
    (copies start without them)."""
    import copy
    from exploring_meta_amd.core_functions.rl import Replay
    r = Replay(states=1, actions=2)
    assert r._mi_pack is None and dict(r) == {'states': 1, 'actions': 2}
    for change in (lambda d: d.__setitem__('states', 3), lambda d: d.update(actions=4), lambda d: d.pop('actions'), lambda d: d.setdefault('x', 0),
                   lambda d: d.__delitem__('x'), lambda d: d.popitem(), lambda d: d.clear(), lambda d: d.__ior__({'y': 1})):
        r._mi_pack = ('memo',)
        change(r)
        assert r._mi_pack is None
    r['states'] = 5
    r._mi_pack = ('memo',)
    assert copy.copy(r)._mi_pack is None and copy.deepcopy(r)._mi_pack is None and dict(copy.deepcopy(r)) == dict(r)
