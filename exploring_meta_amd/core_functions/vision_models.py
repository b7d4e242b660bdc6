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

    def forward(self, x):
        s = self.spec()
        x = x.reshape(-1, s.in_channels, s.in_h, s.in_w).float().contiguous()
        return self.engine().forward_logits(flatten_parameters(self), x.unsqueeze(0))[0]


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
