"""``DiagNormalPolicy`` with the reference's constructor, parameter names/order and initialisers
(core_functions/policies.py:30-67); the MLP runs on the GPU through ``mi_policy_forward``."""
import math

import torch
from torch import nn
from torch.distributions import Normal

from ..engine import PolicyEngine

EPSILON = 1e-6

_engines = {}


def linear_init(module):
    """reference policies.py:17-21"""
    if isinstance(module, nn.Linear):
        nn.init.xavier_uniform_(module.weight)
        module.bias.data.zero_()
    return module


class DiagNormalPolicy(nn.Module):
    def __init__(self, input_size, output_size, hiddens=None, activation='relu'):
        super().__init__()
        if hiddens is None:
            hiddens = [100, 100]
        if activation != 'relu':
            raise ValueError('the HIP policy path implements the ReLU policy (the reference default; rl/maml_trpo.py:86 never '
                             'passes params["activation"])')
        layers = [linear_init(nn.Linear(input_size, hiddens[0])), nn.ReLU()]
        for i, o in zip(hiddens[:-1], hiddens[1:]):
            layers += [linear_init(nn.Linear(i, o)), nn.ReLU()]
        layers.append(linear_init(nn.Linear(hiddens[-1], output_size)))
        self.mean = nn.Sequential(*layers)
        self.sigma = nn.Parameter(torch.Tensor(output_size))
        self.sigma.data.fill_(math.log(1))
        self.input_size, self.output_size, self.hiddens = input_size, output_size, tuple(hiddens)

    def engine(self):
        dev = self.sigma.device
        if dev.type != 'cuda':
            raise RuntimeError('DiagNormalPolicy computes only on the GPU (HIP engine); move it with .to("cuda")')
        key = (self.input_size, self.output_size, self.hiddens, dev.index)
        if key not in _engines:
            _engines[key] = PolicyEngine(self.input_size, self.output_size, self.hiddens, dev)
        return _engines[key]

    def flat(self):
        """Flat parameter vector in named_parameters() order (sigma first)."""
        return torch.cat([p.detach().reshape(-1) for p in self.parameters()]).float().contiguous()

    def load_flat(self, theta):
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(theta[off:off + p.numel()].view_as(p))
                off += p.numel()

    def density(self, state):
        """reference policies.py:49-52"""
        st = state.reshape(1, -1, self.input_size).float().contiguous()
        loc = self.engine().forward(self.flat(), st)[0].reshape(*state.shape[:-1], self.output_size)
        scale = torch.exp(torch.clamp(self.sigma.detach(), min=math.log(EPSILON)))
        return Normal(loc=loc, scale=scale)

    def log_prob(self, state, action):
        """reference policies.py:54-56"""
        return self.density(state).log_prob(action).mean(dim=1, keepdim=True)

    def forward(self, state):
        """reference policies.py:58-61"""
        return self.density(state).sample()
