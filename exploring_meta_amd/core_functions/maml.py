"""``MAML`` wrapper with the call surface of the reference's ``core_functions/maml.py`` (a learn2learn ``MAML`` subclass),
without the learn2learn dependency.

``maml = MAML(model, lr, first_order=False)``; ``learner = maml.clone()``; ``fast_adapt(batch, learner, ...)``;
``eval_loss.backward()`` accumulates into ``maml.parameters()``'s ``.grad`` exactly like the reference loop
(vision/maml_vision.py:84,104-112).  A clone does not copy tensors: it records the base module, and the fused HIP engine
starts every task from the base parameters (the semantics of learn2learn's ``clone_module``).
"""
import torch


class MAML(torch.nn.Module):
    def __init__(self, model, lr, first_order=False, allow_unused=None, allow_nograd=False):
        super().__init__()
        self.module = model
        self.lr = lr
        self.first_order = first_order
        self.allow_nograd = allow_nograd
        self.allow_unused = allow_nograd if allow_unused is None else allow_unused

    def __getattr__(self, attr):
        try:
            return super().__getattr__(attr)
        except AttributeError:
            return getattr(self.__dict__['_modules']['module'], attr)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def clone(self, first_order=None, allow_unused=None, allow_nograd=None):
        """reference core_functions/maml.py:23-49"""
        if first_order is None:
            first_order = self.first_order
        if allow_unused is None:
            allow_unused = self.allow_unused
        if allow_nograd is None:
            allow_nograd = self.allow_nograd
        return MAML(self.module, lr=self.lr, first_order=first_order, allow_unused=allow_unused, allow_nograd=allow_nograd)

    def adapt(self, loss, first_order=None, allow_unused=None, allow_nograd=None):
        raise NotImplementedError(
            'step-wise learner.adapt(loss) is fused into fast_adapt() in this engine (the K inner steps, the query pass and '
            'the second-order outer backward run as one batched HIP call); call core_functions.vision.fast_adapt instead.')

    def get_rep(self, input_d):
        """reference core_functions/maml.py:15-16"""
        return self.get_base_representation(input_d)

    def get_rep_i(self, input_d, layer_i):
        """reference core_functions/maml.py:18-19"""
        return self.get_rep_layer(input_d, layer_i)
