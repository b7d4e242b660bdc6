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


def running_stats_contribution(stats, positions, total, momentum=0.1):
    """Weighted sum of the batch statistics of some of the `total` forward passes of one meta-iteration, [2, C_total]:
    sum_i momentum (1 - momentum)^(total - 1 - positions[i]) stats[i].  torch.nn.BatchNorm2d's running-statistics recurrence
    (running <- (1 - momentum) running + momentum batch, once per forward pass in train mode) is linear, so the buffers after the
    iteration are (1 - momentum)^total running_0 + the SUM of the contributions of all passes -- which lets each rank of a
    task-sharded run fold its own passes and the sum ride in the gradient all-reduce (`apply_running_stats`).

    stats:     [..., 2, C_total] batch mean / biased variance per pass (engine export, `MetaEngine.set_bn_export`).
    positions: [...] index of each pass in the reference's call order (maml_vision.py:102-124: per task, adapt_steps support
               passes and the query pass of the train task, then the same of the validation task)."""
    pos = torch.as_tensor(positions, device=stats.device, dtype=torch.float64)
    w = momentum * (1.0 - momentum) ** (total - 1 - pos)
    return (w.reshape(-1, 1, 1) * stats.reshape(-1, 2, stats.shape[-1]).double()).sum(0).float()


def apply_running_stats(base, spec, contribution, total, images, momentum=0.1):
    """The buffers torch.nn.BatchNorm2d (reference vision_models.py:168-174) leaves after `total` train-mode forward passes of
    `images` images each: learn2learn's clone_module copies parameters but shares buffers, so every `learner(x)` of the reference's
    loop updates the running_mean / running_var / num_batches_tracked that utils/experiment.py:85-90 saves with the model.  The
    running variance uses the UNBIASED batch variance (x n/(n-1), n = images x conv-output pixels of the block); the batch
    statistics themselves never enter the computation (the reference never calls .eval())."""
    from ..utils.roofline import layer_geometry
    blocks = [b for b in base if isinstance(b, ConvBlock)]
    decay = (1.0 - momentum) ** total
    off = 0
    for blk, (_, _, _, co, ho, wo, _, _) in zip(blocks, layer_geometry(spec)):
        n = images * ho * wo
        rm, rv = blk.normalize.running_mean, blk.normalize.running_var
        c = contribution[:, off:off + co].to(rm.device)
        with torch.no_grad():
            rm.mul_(decay).add_(c[0])
            rv.mul_(decay).add_(c[1] * (n / (n - 1.0) if n > 1 else 1.0))
            blk.normalize.num_batches_tracked += total
        off += co


class RunningStatsFold:
    """One meta-iteration's BatchNorm buffer update for a driver (maml_vision.py / anil_vision.py loops): switch the engine's
    export on, `collect(phase)` after each fused call (phase 0 = the train tasks, 1 = the validation tasks the reference adapts
    right after each train task), put `contribution` into the gradient all-reduce and `apply` the reduced sum.

    engine/base/spec: the MetaEngine, the ConvBase holding the buffers and the engine's ModelSpec.
    tasks/lo/hi:      meta_batch_size and this rank's task range.
    passes/images:    forward passes per task per phase (adapt_steps + 1 for MAML, 1 for ANIL) and images per pass."""

    def __init__(self, engine, base, spec, tasks, lo, hi, passes, images, phases=2, momentum=0.1):
        self.engine, self.base, self.spec = engine, base, spec
        self.tasks, self.lo, self.hi, self.passes, self.images, self.phases, self.momentum = tasks, lo, hi, passes, images, phases, momentum
        self.total = tasks * phases * passes
        ctot = spec.hidden * spec.n_layers
        self.contribution = torch.zeros(2, ctot, dtype=torch.float32, device=engine.device)
        self.export = engine.set_bn_export(hi - lo, passes) if hi > lo else None

    def collect(self, phase):
        p = torch.arange(self.passes, dtype=torch.float64).reshape(-1, 1)
        t = torch.arange(self.lo, self.hi, dtype=torch.float64).reshape(1, -1)
        positions = (t * self.phases + phase) * self.passes + p                               # [passes, local tasks]
        self.contribution += running_stats_contribution(self.export, positions, self.total, self.momentum)

    def apply(self, reduced=None):
        self.engine.set_bn_export(0)
        apply_running_stats(self.base, self.spec, self.contribution if reduced is None else reduced, self.total, self.images, self.momentum)


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
    but x and theta) and is itself differentiable (mi_learner_hvp), so `learner.adapt(loss)` of a second-order learner -- learn2learn
    `grad(loss, params, create_graph=True)` -- back-propagates the curvature term.  Training loops should still prefer
    fast_adapt / meta_batch_adapt, which fuse the K inner steps with the second-order backward into one call."""

    @staticmethod
    def forward(ctx, engine, x, theta):
        if x.requires_grad:
            raise RuntimeError('gradients with respect to the input images are not provided by the HIP engine')
        ctx.engine = engine
        ctx.save_for_backward(x, theta)
        return engine.learner_forward(theta.detach().contiguous(), x.unsqueeze(0))[0][0]

    @staticmethod
    def backward(ctx, dlogits):
        x, theta = ctx.saved_tensors
        return None, None, _LearnerBackward.apply(ctx.engine, x, theta, dlogits)


class _LearnerBackward(torch.autograd.Function):
    """g = d sum(logits * dlogits) / d theta (mi_learner_backward) as a differentiable function of (theta, dlogits)."""

    @staticmethod
    def forward(ctx, engine, x, theta, dlogits):
        ctx.engine = engine
        ctx.save_for_backward(x, theta, dlogits)
        return engine.learner_backward(theta.detach().contiguous(), x.unsqueeze(0), dlogits.detach().unsqueeze(0))[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, theta, dlogits = ctx.saved_tensors
        gtheta, ldot = ctx.engine.learner_hvp(theta.detach().contiguous(), x.unsqueeze(0), dlogits.detach().unsqueeze(0), v)
        return None, None, gtheta[0], ldot[0]


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
