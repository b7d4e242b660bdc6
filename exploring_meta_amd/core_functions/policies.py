"""``DiagNormalPolicy`` / ``DiagNormalPolicyANIL`` with the reference's constructors, parameter names/order and initialisers
(core_functions/policies.py:30-67,70-126); the MLP runs on the GPU through ``mi_policy_forward``."""
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
        if activation == 'relu':                                              # reference policies.py:32-37
            act = nn.ReLU
        elif activation == 'tanh':
            act = nn.Tanh
        else:
            raise NotImplementedError
        layers = [linear_init(nn.Linear(input_size, hiddens[0])), act()]
        for i, o in zip(hiddens[:-1], hiddens[1:]):
            layers += [linear_init(nn.Linear(i, o)), act()]
        layers.append(linear_init(nn.Linear(hiddens[-1], output_size)))
        self.sigma = nn.Parameter(torch.Tensor(output_size))                  # registered first (named_parameters order)
        self.sigma.data.fill_(math.log(1))
        self.mean = nn.Sequential(*layers)
        self.input_size, self.output_size, self.hiddens, self.activation = input_size, output_size, tuple(hiddens), activation

    def _engine_params(self):
        """Parameters in the engine's order: sigma, then (weight, bias) of the three Linear layers."""
        lin = [m for m in self.mean if isinstance(m, nn.Linear)]
        return [self.sigma] + [q for m in lin for q in (m.weight, m.bias)]

    def engine(self):
        dev = self.sigma.device
        if dev.type != 'cuda':
            raise RuntimeError(f'{type(self).__name__} computes only on the GPU (HIP engine); move it with .to("cuda")')
        key = (self.input_size, self.output_size, self.hiddens, self.activation, dev.index)
        if key not in _engines:
            _engines[key] = PolicyEngine(self.input_size, self.output_size, self.hiddens, dev, activation=self.activation)
        return _engines[key]

    def flat(self):
        """Flat parameter vector in the engine's order (sigma, W1, b1, W2, b2, W3, b3)."""
        return torch.cat([p.detach().reshape(-1) for p in self._engine_params()]).float().contiguous()

    def load_flat(self, theta):
        params = self._engine_params()
        f32 = torch.float32
        if (theta.is_cuda and theta.dtype is f32 and theta.is_contiguous() and
                all(p.is_cuda and p.dtype is f32 and p.is_contiguous() and p.get_device() == theta.get_device() for p in params)):
            # one mi_copy_segments launch scatters the vector into the parameter tensors (one copy launch per parameter otherwise)
            from ..engine import copy_segments
            base, srcs, cnts, off = theta.data_ptr(), [], [], 0
            for p in params:
                srcs.append(base + 4 * off)
                cnts.append(p.numel())
                off += p.numel()
            if off != theta.numel():
                raise ValueError(f'load_flat: {theta.numel()} values for {off} parameters')
            copy_segments(srcs, [p.data_ptr() for p in params], cnts, cnts, theta.device)      # (like a write through `.data`: no autograd version bump)
            return
        off = 0
        with torch.no_grad():
            for p in params:
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


class DiagNormalPolicyANIL(DiagNormalPolicy):
    """reference policies.py:70-126: tanh body (`self.body`), linear head (`self.head`), `sigma` registered last;
    `turn_off_body_grads()` puts the body under no_grad for the inner loop (rl.py:381-382), which on this engine is the
    `head_only` mode of mi_policy_adapt."""

    def __init__(self, input_size, output_size, fc_neurons, hiddens=None):
        nn.Module.__init__(self)
        if hiddens is None:
            hiddens = [100, 100]
        if len(hiddens) != 2 or fc_neurons != hiddens[-1]:
            raise ValueError('fc_neurons must equal the width of the last hidden layer (head = Linear(fc_neurons, output_size))')
        self.fc_neurons = fc_neurons
        layers = [linear_init(nn.Linear(input_size, hiddens[0])), nn.Tanh()]
        for i, o in zip(hiddens[:-1], hiddens[1:]):
            layers += [linear_init(nn.Linear(i, o)), nn.Tanh()]
        self.body = nn.Sequential(*layers)
        self.head = linear_init(nn.Linear(fc_neurons, output_size))
        self.sigma = nn.Parameter(torch.Tensor(output_size))
        self.sigma.data.fill_(math.log(1))
        self.features_no_grad = False
        self.input_size, self.output_size, self.hiddens, self.activation = input_size, output_size, tuple(hiddens), 'tanh'

    def _engine_params(self):
        lin = [m for m in self.body if isinstance(m, nn.Linear)] + [self.head]
        return [self.sigma] + [q for m in lin for q in (m.weight, m.bias)]

    def turn_on_body_grads(self):
        self.features_no_grad = False

    def turn_off_body_grads(self):
        self.features_no_grad = True
