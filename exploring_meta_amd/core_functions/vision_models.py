"""The reference's few-shot classifiers (core_functions/vision_models.py) as parameter containers for the HIP engine.

Same class names, constructor arguments, parameter registration order, state_dict keys (so a reference ``model.pt`` loads
with ``load_state_dict``) and initialisers as the reference; ``forward`` runs on the GPU through ``mi_forward_logits``
(BatchNorm always in train mode, like the reference which never calls ``.eval()``).  There is no CPU forward here.
"""
import torch

from ..engine import MetaEngine, ModelSpec, flatten_parameters


def maml_init_(module):
    """reference vision_models.py:204-207"""
    torch.nn.init.xavier_uniform_(module.weight.data, gain=1.0)
    torch.nn.init.constant_(module.bias.data, 0.0)
    return module


class _Params(torch.nn.Module):
    """A module that only holds parameters/buffers under the reference's attribute names."""


class ConvBlock(torch.nn.Module):
    """reference vision_models.py:149-193 (conv 3x3 pad 1 -> BatchNorm2d(train) -> ReLU -> MaxPool2d / identity)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), max_pool=True, max_pool_factor=1.0):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or int(2 * max_pool_factor) != 2:
            raise ValueError('the HIP engine implements the reference configuration: 3x3 kernels, pooling/stride factor 2')
        self.max_pool = bool(max_pool)
        self.normalize = _Params()
        self.normalize.weight = torch.nn.Parameter(torch.empty(out_channels))
        self.normalize.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.normalize.register_buffer('running_mean', torch.zeros(out_channels))
        self.normalize.register_buffer('running_var', torch.ones(out_channels))
        self.normalize.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        torch.nn.init.uniform_(self.normalize.weight)                      # reference :175
        self.conv = _Params()
        self.conv.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.conv.bias = torch.nn.Parameter(torch.empty(out_channels))
        maml_init_(self.conv)                                              # reference :186


class ConvBase(torch.nn.Sequential):
    """reference vision_models.py:121-146"""

    def __init__(self, output_size, hidden=64, channels=1, max_pool=False, layers=4, max_pool_factor=1.0):
        core = [ConvBlock(channels, hidden, (3, 3), max_pool=max_pool, max_pool_factor=max_pool_factor)]
        for _ in range(layers - 1):
            core.append(ConvBlock(hidden, hidden, kernel_size=(3, 3), max_pool=max_pool, max_pool_factor=max_pool_factor))
        super().__init__(*core)
        self.hidden, self.channels, self.max_pool, self.layers = hidden, channels, bool(max_pool), layers


class _EngineModel(torch.nn.Module):
    _engines = {}

    def spec(self):
        raise NotImplementedError

    def engine(self):
        """One MetaEngine per (architecture, device), shared by all instances/clones."""
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('this model computes only on the GPU (HIP engine); move it with .to("cuda")')
        key = (self.spec(), dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in _EngineModel._engines:
            _EngineModel._engines[key] = MetaEngine(self.spec(), dev)
        return _EngineModel._engines[key]

    def flat_parameters(self):
        """parameters() as one flat tensor that is still connected to them in the autograd graph."""
        return torch.cat([p.reshape(-1) for p in self.parameters()]).float()

    def _images(self, x):
        s = self.spec()
        return x.reshape(-1, s.in_channels, s.in_h, s.in_w).float().contiguous()

    def forward(self, x, theta=None):
        """`model(x)` (reference vision_models.py:51-55,107-110).  `theta` (flat, parameters() order) overrides the module's
        own parameters -- a learner's fast weights.  Differentiable once w.r.t. the parameters (mi_learner_backward)."""
        theta = self.flat_parameters() if theta is None else theta
        return _LearnerForward.apply(self.engine(), self._images(x), theta)

    def get_base_representation(self, x, theta=None):
        """reference vision_models.py:57-58,112-113: `self.base(x)`, NCHW [N, hidden, h, w]."""
        return self.get_rep_layer(x, self.layers, theta)

    def get_rep_layer(self, x, layer, theta=None):
        """reference vision_models.py:60-63,115-118: layer == -1 -> `self.linear(x.view(-1, 25 * hidden))` on a given base
        representation; otherwise the output of the first `layer` ConvBlocks (0 = x itself)."""
        theta = (self.flat_parameters() if theta is None else theta).detach()
        if layer == -1:
            f = x.reshape(-1, 25 * self.hidden_size).float().contiguous()     # raises like the reference's view for Omniglot
            nl = self.linear.weight.numel() + self.linear.bias.numel()
            wl = theta[-nl:-self.linear.bias.numel()].reshape(self.linear.weight.shape)
            return self.engine().head_logits(f, wl, theta[-self.linear.bias.numel():])
        if layer == 0:
            return x
        if not 1 <= layer <= self.layers:
            raise ValueError(f'layer must be -1 or in 0..{self.layers}')
        return self.engine().learner_forward(theta, self._images(x).unsqueeze(0), rep_layer=layer, want_logits=False)[1][0]


class _LearnerForward(torch.autograd.Function):
    """logits = net(x; theta) through mi_learner_forward; backward = mi_learner_backward (re-runs the forward, keeps nothing
    but x and theta).  Once differentiable: a second-order meta-gradient through step-wise calls raises -- use fast_adapt /
    meta_batch_adapt, which fuse the K inner steps with the second-order backward."""

    @staticmethod
    def forward(ctx, engine, x, theta):
        if x.requires_grad:
            raise RuntimeError('gradients with respect to the input images are not provided by the HIP engine')
        theta_d = theta.detach().contiguous()
        ctx.engine = engine
        ctx.save_for_backward(x, theta_d)
        return engine.learner_forward(theta_d, x.unsqueeze(0))[0][0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        x, theta = ctx.saved_tensors
        return None, None, ctx.engine.learner_backward(theta, x.unsqueeze(0), dlogits.unsqueeze(0))[0]


class MiniImagenetCNN(_EngineModel):
    """reference vision_models.py:66-118"""

    def __init__(self, output_size, hidden_size=32, layers=4):
        super().__init__()
        if layers != 4:
            raise ValueError('max_pool_factor = 4 // layers must be 1 (layers=4) for the HIP engine')
        self.base = ConvBase(output_size=hidden_size, hidden=hidden_size, channels=3, max_pool=True, layers=layers,
                             max_pool_factor=4 // layers)
        self.linear = torch.nn.Linear(25 * hidden_size, output_size, bias=True)
        maml_init_(self.linear)
        self.hidden_size, self.output_size, self.layers = hidden_size, output_size, layers

    def spec(self):
        return ModelSpec.mini_imagenet(self.output_size, self.hidden_size, self.layers)


class OmniglotCNN(_EngineModel):
    """reference vision_models.py:10-63"""

    def __init__(self, output_size=5, hidden_size=64, layers=4):
        super().__init__()
        self.hidden_size, self.output_size, self.layers = hidden_size, output_size, layers
        self.base = ConvBase(output_size=hidden_size, hidden=hidden_size, channels=1, max_pool=False, layers=layers)
        self.linear = torch.nn.Linear(hidden_size, output_size, bias=True)
        self.linear.weight.data.normal_()                                  # reference :48-49
        self.linear.bias.data.mul_(0.0)

    def spec(self):
        return ModelSpec.omniglot(self.output_size, self.hidden_size, self.layers)
