"""``MAML`` wrapper with the call surface of the reference's ``core_functions/maml.py`` (a learn2learn ``MAML`` subclass),
without the learn2learn dependency.

``maml = MAML(model, lr, first_order=False)``; ``learner = maml.clone()``; ``fast_adapt(batch, learner, ...)``;
``eval_loss.backward()`` accumulates into ``maml.parameters()``'s ``.grad`` exactly like the reference loop
(vision/maml_vision.py:84,104-112).  A clone does not copy tensors: it records the base module, and the fused HIP engine
starts every task from the base parameters (the semantics of learn2learn's ``clone_module``).

The step-wise surface (``learner(x)``, ``learner.adapt(loss)``, ``get_rep``, ``get_rep_i`` -- what the reference's
misc_scripts/cl_vision.py and rc_vision.py drive) holds the fast weights as one flat tensor and runs every forward /
gradient through ``mi_learner_forward`` / ``mi_learner_backward``.
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
        self.__dict__['_fast'] = None      # flat fast weights of a step-wise learner (plain tensor, not a Parameter)

    def __getattr__(self, attr):
        try:
            return super().__getattr__(attr)
        except AttributeError:
            return getattr(self.__dict__['_modules']['module'], attr)

    def fast_weights(self):
        """Flat fast weights (parameters() order), connected to the base parameters in the autograd graph."""
        if self.__dict__['_fast'] is None:
            self.__dict__['_fast'] = self.module.flat_parameters()
        return self.__dict__['_fast']

    def forward(self, x):
        """`learner(x)`: the module evaluated with this learner's current fast weights (mi_learner_forward)."""
        if not hasattr(self.module, 'flat_parameters'):
            return self.module(x)
        return self.module(x, theta=self.fast_weights())

    def clone(self, first_order=None, allow_unused=None, allow_nograd=None):
        """reference core_functions/maml.py:23-49 (learn2learn clone_module: the clone starts from the CURRENT weights of
        this learner and stays connected to them in the graph)."""
        if first_order is None:
            first_order = self.first_order
        if allow_unused is None:
            allow_unused = self.allow_unused
        if allow_nograd is None:
            allow_nograd = self.allow_nograd
        c = MAML(self.module, lr=self.lr, first_order=first_order, allow_unused=allow_unused, allow_nograd=allow_nograd)
        c.__dict__['_fast'] = self.__dict__['_fast']
        return c

    def adapt(self, loss, first_order=None, allow_unused=None, allow_nograd=None):
        """learn2learn `MAML.adapt` (call sites core_functions/vision.py:13, misc_scripts/cl_vision.py:59,
        rc_vision.py:70): g = grad(loss, fast weights); p <- p - lr * g, out of place.

        `loss` must come from `learner(x)` of this learner.  The gradient is one mi_learner_backward call.  The update keeps the
        identity path to the base parameters; a first-order learner detaches the gradient (first-order meta-gradient), a
        second-order one keeps it in the graph (create_graph) and a later `.backward()` differentiates through it with
        mi_learner_hvp -- the exact second-order meta-gradient, one Hessian-vector sweep per adapt step.  Training loops are
        faster through `fast_adapt` / `meta_batch_adapt` (one fused HIP call for the K steps, the query pass and the outer
        backward for a whole meta-batch)."""
        if first_order is None:
            first_order = self.first_order
        second_order = not first_order
        theta = self.fast_weights()
        if not theta.requires_grad:
            raise RuntimeError('learner.adapt needs parameters that require grad')
        (g,) = torch.autograd.grad(loss, theta, retain_graph=second_order, create_graph=second_order)
        self.__dict__['_fast'] = theta - self.lr * g

    def get_rep(self, input_d):
        """reference core_functions/maml.py:15-16"""
        return self.module.get_base_representation(input_d, theta=self.fast_weights())

    def get_rep_i(self, input_d, layer_i):
        """reference core_functions/maml.py:18-19"""
        return self.module.get_rep_layer(input_d, layer_i, theta=self.fast_weights())
