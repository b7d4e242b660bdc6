"""Batched MAML engine: Python host over the C ABI (include/mi_maml.h).

``MetaEngine.meta_batch`` processes a whole meta-batch of tasks in one call -- what the reference does with a sequential
Python loop of ``maml.clone()`` / ``fast_adapt`` / ``eval_loss.backward()`` (vision/maml_vision.py:102-114).  PyTorch is
used only for device memory and streams; all arithmetic runs in libmi_maml's HIP kernels.
"""

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib


@dataclass(frozen=True)
class ModelSpec:
    """Architecture of the reference's few-shot classifiers (core_functions/vision_models.py)."""
    n_layers: int
    in_channels: int
    in_h: int
    in_w: int
    hidden: int
    max_pool: bool
    ways: int
    head_mean_pool: bool

    @staticmethod
    def mini_imagenet(ways, hidden=32, layers=4):
        """MiniImagenetCNN(output_size=ways, hidden_size=32, layers=4) (vision_models.py:93-105)."""
        return ModelSpec(layers, 3, 84, 84, hidden, True, ways, False)

    @staticmethod
    def omniglot(ways, hidden=64, layers=4):
        """OmniglotCNN(output_size=ways, hidden_size=64, layers=4) (vision_models.py:39-49)."""
        return ModelSpec(layers, 1, 28, 28, hidden, False, ways, True)

    @staticmethod
    def anil(ways, hidden=64, channels=3, max_pool=True, layers=4, in_hw=84):
        """ANIL trunk ConvBase(output_size, hidden, channels, max_pool) + Linear(fc_neurons, ways) head
        (vision/anil_vision.py:86-94): Mini-ImageNet (64 filters, 3x84x84, pooling) or Omniglot (32 filters, 1x28x28)."""
        return ModelSpec(layers, channels, in_hw, in_hw, hidden, bool(max_pool), ways, False)

    def block_output_shape(self, layer):
        """(C, H, W) after the first `layer` ConvBlocks (layer in 1..n_layers)."""
        if not 1 <= layer <= self.n_layers:
            raise ValueError(f'layer must be in 1..{self.n_layers}')
        h, w = self.in_h, self.in_w
        for _ in range(layer):
            if self.max_pool:
                h, w = h // 2, w // 2
            else:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return self.hidden, h, w

    def param_shapes(self):
        """(name, shape) in the reference's parameters() order / state_dict naming."""
        out, ci = [], self.in_channels
        for i in range(self.n_layers):
            out += [(f'base.{i}.normalize.weight', (self.hidden,)), (f'base.{i}.normalize.bias', (self.hidden,)),
                    (f'base.{i}.conv.weight', (self.hidden, ci, 3, 3)), (f'base.{i}.conv.bias', (self.hidden,))]
            ci = self.hidden
        h, w = self.in_h, self.in_w
        for _ in range(self.n_layers):
            if self.max_pool:
                h, w = h // 2, w // 2
            else:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        feat = self.hidden if self.head_mean_pool else self.hidden * h * w
        out += [('linear.weight', (self.ways, feat)), ('linear.bias', (self.ways,))]
        return out


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _on_device(fn):
    """Run an engine method with the engine's own device current (launches go to that device's current stream), whatever
    device the caller has selected."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapped


def _resolve_device(device):
    if device is None:
        return torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.type != 'cuda':
        raise _lib.MiError(f'the engine runs on a GPU, got device {device}')
    return torch.device('cuda', device.index if device.index is not None else torch.cuda.current_device())


class MetaEngine:
    """One engine per (model spec, device).  Not re-entrant (one host thread per GPU / rank)."""

    def __init__(self, spec, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MiError('MetaEngine needs a GPU: the MAML hot path has no CPU implementation in this package '
                               '(the CPU restatement under oracle/ is test infrastructure only).')
        self.spec = spec
        self.device = _resolve_device(device)
        desc = _lib.MiModelDesc(spec.n_layers, spec.in_channels, spec.in_h, spec.in_w, spec.hidden, int(spec.max_pool),
                                spec.ways, int(spec.head_mean_pool))
        self._h = C.c_void_p()
        _lib.check(self.lib.mi_engine_create(C.byref(desc), self.device.index, C.byref(self._h)))
        n = C.c_size_t()
        _lib.check(self.lib.mi_param_count(self._h, C.byref(n)), self._h)
        self.param_count = n.value
        self._ws = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.mi_engine_destroy(h)
            self._h = C.c_void_p()

    def set_fused_block1(self, on):
        """Ablation/test switch for the conv-recompute kernels of block 1: 0 generic kernels, 1 (default) fused kernels with the Gram-matrix path for
        the support passes and the query pass, 2 fused kernels without it, 3 the Gram-matrix path for the support passes only."""
        _lib.check(self.lib.mi_engine_set_fused_block1(self._h, int(on)), self._h)

    def set_overlap(self, on):
        """Side-stream execution of the weight gradients of blocks >= 2 (default on: one fork of the side stream per hidden block; 3: one
        per backward pass, measured slower); results do not depend on it."""
        _lib.check(self.lib.mi_engine_set_overlap(self._h, int(on)), self._h)

    def set_fused_finalize(self, on):
        """BatchNorm partials folded inside the producing kernels (default on) or by separate launches; bit-identical results."""
        _lib.check(self.lib.mi_engine_set_fused_finalize(self._h, int(on)), self._h)

    def set_fused_block1_reduce(self, on):
        """Block 1's BatchNorm-backward sums in the epilogue of block 2's dgrad (default on) or as a separate streaming pass."""
        _lib.check(self.lib.mi_engine_set_fused_block1_reduce(self._h, int(on)), self._h)

    def set_fused_tail(self, on):
        """One "advance" launch at the end of every pass of `meta_batch` (default on) or the separate fold / update / statistics
        launches; bit-identical results."""
        _lib.check(self.lib.mi_engine_set_fused_tail(self._h, int(on)), self._h)

    def set_fused_last_block(self, on):
        """The last block's BatchNorm + pooling, the head, its backward and that block's BatchNorm backward (or their tangents) as one launch
        per pass with one workgroup per task (default on) or the five separate launches (mi_engine_set_fused_last_block)."""
        _lib.check(self.lib.mi_engine_set_fused_last_block(self._h, int(on)), self._h)

    def set_graph(self, on):
        """Replay repeated identical fused calls as one hipGraphLaunch (mi_engine_set_graph).  While on, `meta_batch` /
        `meta_batch_anil` return views of PERSISTENT output buffers (one set per call shape), because a replay writes where the
        captured call wrote: results are valid until the next call of the same shape; copy what must outlive it.  The caller keeps
        theta / data / labels in place between iterations (in-place optimizer step, resident task batches) to benefit."""
        _lib.check(self.lib.mi_engine_set_graph(self._h, int(on)), self._h)
        self._graph = bool(on)
        self._persist = {}
        # stream capture is not permitted on the legacy default stream torch hands out as "current": graph-mode calls run on an
        # engine-owned stream, ordered after / before the caller's current stream with events
        self._gstream = torch.cuda.Stream(device=self.device) if on else None

    def _fused_call(self, fn, *args):
        """Issue one fused C call on the caller's current stream, or (graph mode) on the engine's capture-capable stream, fenced
        against the current stream on both sides."""
        gs = getattr(self, '_gstream', None)
        if gs is None:
            return fn(self._h, _stream(self.device), *args)
        cur = torch.cuda.current_stream(self.device)
        gs.wait_stream(cur)
        rc = fn(self._h, C.c_void_p(gs.cuda_stream), *args)
        cur.wait_stream(gs)
        return rc

    def _outputs(self, kind, T, nlog, with_grad, return_logits):
        """(loss, acc, grad, logits) buffers of one fused call: fresh tensors, or the persistent set of this shape in graph mode."""
        def fresh():
            # one allocation [grad (P) | loss (T) | acc (T)]: a data-parallel caller all-reduces the three as ONE contiguous buffer
            # (sharding.MetaTrainer) without gathering them into a new tensor first
            P = self.param_count if with_grad else 0
            buf = torch.empty(P + 2 * T, dtype=torch.float32, device=self.device)
            return (buf[P:P + T], buf[P + T:], buf[:P] if with_grad else None,
                    torch.empty(T, nlog, self.spec.ways, dtype=torch.float32, device=self.device) if return_logits else None)
        if not getattr(self, '_graph', False):
            return fresh()
        key = (kind, T, nlog, bool(with_grad), bool(return_logits))
        if key not in self._persist:
            self._persist[key] = fresh()
        return self._persist[key]

    def set_bn_export(self, tasks=0, passes=0):
        """mi_engine_set_bn_export: while on, every fused call also leaves the BatchNorm batch statistics of each of its forward
        passes in the returned tensor [passes, tasks, 2, C_total] (mean, biased variance; C_total = blocks x filters, block-major).
        passes = adapt_steps + 1 for meta_batch (inner steps, then the query pass), 1 for meta_batch_anil.  tasks = 0 switches it off."""
        if not tasks:
            _lib.check(self.lib.mi_engine_set_bn_export(self._h, C.c_void_p(0), 0), self._h)
            self._bn_export = None
            return None
        ctot = self.spec.hidden * self.spec.n_layers
        self._bn_export = torch.zeros(passes, tasks, 2, ctot, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_engine_set_bn_export(self._h, _ptr(self._bn_export), self._bn_export.numel()), self._h)
        return self._bn_export

    def set_trace(self, tasks=0, adapt_steps=0):
        """Debug/test aid (mi_debug_set_trace): allocate a trace buffer for meta_batch calls with these sizes and return it as a
        dict of views {theta [K+1,T,P], g [K,T,P], lam_in [K,T,P], hv [K,T,P]} (reference parameter order); 0 tasks switches it off."""
        if not tasks:
            _lib.check(self.lib.mi_debug_set_trace(self._h, None, 0), self._h)
            self._trace = None
            return None
        K, T, P = adapt_steps, tasks, self.param_count
        buf = torch.zeros((4 * K + 1) * T * P, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_debug_set_trace(self._h, _ptr(buf), buf.numel()), self._h)
        self._trace = buf
        v = buf.view(4 * K + 1, T, P)
        return dict(theta=v[:K + 1], g=v[K + 1:2 * K + 1], lam_in=v[2 * K + 1:3 * K + 1], hv=v[3 * K + 1:])

    def workspace_bytes(self, tasks, shots, adapt_steps, second_order):
        b = C.c_size_t()
        _lib.check(self.lib.mi_workspace_bytes(self._h, tasks, self.spec.ways, shots, adapt_steps, int(second_order),
                                               C.byref(b)), self._h)
        return b.value

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    @_on_device
    def meta_batch(self, theta, data, labels, shots, adapt_steps, inner_lr, first_order=False, with_grad=True,
                   return_logits=False, grad_tasks=None):
        """theta [P] fp32; data [T, 2*shots*ways, C, H, W] fp32 (the reference's task batches, stacked); labels [T, 2*S*W]
        int64.  Returns (loss[T], acc[T], meta_grad[P] summed over tasks or None, logits [T, S*W, ways] or None).
        grad_tasks = G (0 < G < T): the first G tasks are the meta-iteration's TRAIN tasks (the meta-gradient is summed over them), the
        other T - G its VALIDATION tasks, adapted and scored in the same launches without a backward half (maml_vision.py:117-124)."""
        s = self.spec
        T = data.shape[0]
        n2 = 2 * shots * s.ways
        if tuple(data.shape) != (T, n2, s.in_channels, s.in_h, s.in_w) and \
                not (s.in_channels == 1 and data.numel() == T * n2 * s.in_h * s.in_w):
            raise ValueError(f'data shape {tuple(data.shape)} does not match [T, {n2}, {s.in_channels}, {s.in_h}, {s.in_w}]')
        if tuple(labels.shape) != (T, n2):
            raise ValueError(f'labels shape {tuple(labels.shape)} != {(T, n2)}')
        if theta.numel() != self.param_count:
            raise ValueError(f'theta has {theta.numel()} elements, the model has {self.param_count}')
        for t, dt in ((theta, torch.float32), (data, torch.float32), (labels, torch.int64)):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise ValueError('theta/data must be contiguous fp32 CUDA tensors and labels contiguous int64 CUDA')
        if grad_tasks is None:
            grad_tasks = T if with_grad else 0
        if not 0 <= grad_tasks <= T:
            raise ValueError(f'grad_tasks must be in 0..{T}, got {grad_tasks}')
        with_grad = grad_tasks > 0
        so = (not first_order) and with_grad
        ws = self._workspace(self.workspace_bytes(T, shots, adapt_steps, so))
        loss, acc, grad, logits = self._outputs('maml', T, shots * s.ways, with_grad, return_logits)
        rc = self._fused_call(self.lib.mi_meta_batch_maml_tv, _ptr(theta), _ptr(data), _ptr(labels), T, int(grad_tasks), s.ways, shots,
                              adapt_steps, float(inner_lr), int(not first_order), _ptr(loss),
                              _ptr(acc), _ptr(grad), _ptr(logits), _ptr(ws), ws.numel())
        _lib.check(rc, self._h)
        return loss, acc, grad, logits

    @_on_device
    def meta_batch_anil(self, theta, data, labels, shots, adapt_steps, inner_lr, first_order=False, with_grad=True,
                        return_logits=False):
        """ANIL (reference vision/anil_vision.py:116-122): theta = [features.parameters()..., head.weight, head.bias] flat;
        the trunk sees all 2*shots*ways images of a task at once, only the head is adapted.  Same returns as meta_batch."""
        s = self.spec
        T = data.shape[0]
        n2 = 2 * shots * s.ways
        if tuple(data.shape) != (T, n2, s.in_channels, s.in_h, s.in_w) or tuple(labels.shape) != (T, n2):
            raise ValueError(f'data/labels shapes {tuple(data.shape)}/{tuple(labels.shape)} do not match T x {n2} task rows')
        if theta.numel() != self.param_count:
            raise ValueError(f'theta has {theta.numel()} elements, trunk + head have {self.param_count}')
        for t, dt in ((theta, torch.float32), (data, torch.float32), (labels, torch.int64)):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise ValueError('theta/data must be contiguous fp32 CUDA tensors and labels contiguous int64 CUDA')
        b = C.c_size_t()
        _lib.check(self.lib.mi_anil_workspace_bytes(self._h, T, s.ways, shots, adapt_steps, C.byref(b)), self._h)
        ws = self._workspace(b.value)
        loss, acc, grad, logits = self._outputs('anil', T, shots * s.ways, with_grad, return_logits)
        rc = self._fused_call(self.lib.mi_meta_batch_anil, _ptr(theta), _ptr(data), _ptr(labels), T, s.ways, shots,
                              adapt_steps, float(inner_lr), int(not first_order), int(with_grad), _ptr(loss),
                              _ptr(acc), _ptr(grad), _ptr(logits), _ptr(ws), ws.numel())
        _lib.check(rc, self._h)
        return loss, acc, grad, logits

    @_on_device
    def forward_logits(self, theta, x):
        """Plain forward (no adaptation): x [T, n, C, H, W] -> logits [T, n, ways]; BatchNorm uses each batch's statistics."""
        T, n = x.shape[0], x.shape[1]
        b = C.c_size_t()
        _lib.check(self.lib.mi_forward_workspace_bytes(self._h, T, n, C.byref(b)), self._h)
        ws = self._workspace(b.value)
        logits = torch.empty(T, n, self.spec.ways, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_forward_logits(self._h, _stream(self.device), _ptr(theta.contiguous()), _ptr(x.contiguous()), T, n,
                                              _ptr(logits), _ptr(ws), ws.numel()), self._h)
        return logits

    def _learner_args(self, theta, x):
        theta = theta.reshape(-1, self.param_count) if theta.dim() == 1 else theta
        if theta.dim() != 2 or theta.shape[1] != self.param_count or theta.shape[0] not in (1, x.shape[0]):
            raise ValueError(f'theta must be [P] or [1 | tasks, P] with P={self.param_count}, got {tuple(theta.shape)}')
        for t in (theta, x):
            if t.dtype != torch.float32 or not t.is_cuda:
                raise ValueError('theta / x must be fp32 CUDA tensors')
        if x.dim() != 5 or tuple(x.shape[2:]) != (self.spec.in_channels, self.spec.in_h, self.spec.in_w):
            raise ValueError(f'x must be [tasks, n, {self.spec.in_channels}, {self.spec.in_h}, {self.spec.in_w}], got {tuple(x.shape)}')
        T, n = x.shape[0], x.shape[1]
        b = C.c_size_t()
        _lib.check(self.lib.mi_forward_workspace_bytes(self._h, T, n, C.byref(b)), self._h)
        return theta.contiguous(), x.contiguous(), T, n, self._workspace(b.value)

    @_on_device
    def learner_forward(self, theta, x, rep_layer=None, want_logits=True):
        """`learner(x)` with caller-held fast weights: theta [P] / [1,P] (shared) or [T,P]; x [T, n, C, H, W].
        Returns (logits [T,n,ways] or None, rep or None) where rep = output of the first `rep_layer` ConvBlocks in NCHW
        (reference get_rep_layer, vision_models.py:60-63,115-118)."""
        theta, x, T, n, ws = self._learner_args(theta, x)
        logits = torch.empty(T, n, self.spec.ways, dtype=torch.float32, device=self.device) if want_logits else None
        rep = None
        if rep_layer is not None:
            c, h, w = self.spec.block_output_shape(rep_layer)
            rep = torch.empty(T, n, c, h, w, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_learner_forward(self._h, _stream(self.device), _ptr(theta), theta.shape[0], _ptr(x), T, n, _ptr(logits),
                                               int(rep_layer or 0), _ptr(rep), _ptr(ws), ws.numel()), self._h)
        return logits, rep

    @_on_device
    def learner_backward(self, theta, x, dlogits):
        """Vector-Jacobian product of `learner(x)`: d sum(logits*dlogits)/d theta, [theta_tasks, P] (first derivatives)."""
        theta, x, T, n, ws = self._learner_args(theta, x)
        dlogits = dlogits.to(torch.float32).contiguous()
        if tuple(dlogits.shape) != (T, n, self.spec.ways):
            raise ValueError(f'dlogits must be {(T, n, self.spec.ways)}, got {tuple(dlogits.shape)}')
        grad = torch.empty(theta.shape[0], self.param_count, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_learner_backward(self._h, _stream(self.device), _ptr(theta), theta.shape[0], _ptr(x), _ptr(dlogits), T, n,
                                                _ptr(grad), _ptr(ws), ws.numel()), self._h)
        return grad

    @_on_device
    def learner_hvp(self, theta, x, dlogits, v):
        """mi_learner_hvp: the vector-Jacobian products of (theta, dlogits) -> learner_backward(theta, x, dlogits) for a cotangent v
        [theta_tasks, P] on that gradient: ((d^2 sum(logits*dlogits)/dtheta^2) v  [theta_tasks, P],  J(theta) v  [T, n, ways])."""
        theta, x, T, n, _ = self._learner_args(theta, x)
        dlogits = dlogits.to(torch.float32).contiguous()
        v = v.to(torch.float32).reshape(theta.shape).contiguous()
        if tuple(dlogits.shape) != (T, n, self.spec.ways):
            raise ValueError(f'dlogits must be {(T, n, self.spec.ways)}, got {tuple(dlogits.shape)}')
        b = C.c_size_t()
        _lib.check(self.lib.mi_learner_hvp_workspace_bytes(self._h, T, n, C.byref(b)), self._h)
        ws = self._workspace(b.value)
        gtheta = torch.empty(theta.shape[0], self.param_count, dtype=torch.float32, device=self.device)
        ldot = torch.empty(T, n, self.spec.ways, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_learner_hvp(self._h, _stream(self.device), _ptr(theta), theta.shape[0], _ptr(x), _ptr(dlogits), _ptr(v), T, n,
                                           _ptr(gtheta), _ptr(ldot), _ptr(ws), ws.numel()), self._h)
        return gtheta, ldot

    @_on_device
    def head_logits(self, f, wl, bl):
        """`linear(f)` for f [n, F] with explicit weights (reference get_rep_layer(x, -1)); mi_head_fwd_bwd, forward only."""
        f, wl, bl = f.contiguous(), wl.contiguous().float(), bl.contiguous().float()
        n, feat, ways = f.shape[0], f.shape[1], wl.shape[0]
        y = torch.zeros(n, dtype=torch.int32, device=self.device)
        scr = torch.empty(max(2 * n, n * feat), dtype=torch.float32, device=self.device)
        la = torch.empty(2, dtype=torch.float32, device=self.device)
        logits = torch.empty(n, ways, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mi_head_fwd_bwd(_stream(self.device), _ptr(f), _ptr(wl), _ptr(bl), 0, _ptr(y), 1, n, feat, ways, _ptr(la[:1]),
                                            _ptr(la[1:]), _ptr(logits), None, None, None, None, 0, _ptr(scr)), self._h)
        return logits

    def profile(self, on, op=None, layer=0):
        """Record HIP events around every launch (or only launches of one (op name, layer)) until switched off."""
        kind = -1
        if op is not None:
            names = [self.lib.mi_profile_op_name(i).decode() for i in range(self.lib.mi_profile_kinds() // 8)]
            kind = names.index(op) * 8 + layer
        _lib.check(self.lib.mi_profile_enable(self._h, int(on), kind), self._h)

    def profile_collect(self):
        """{(op name, layer): (total_ms, launches)} since the last collect; synchronises on the recorded events."""
        n = self.lib.mi_profile_kinds()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(self.lib.mi_profile_collect(self._h, ms, cnt, n), self._h)
        return {(self.lib.mi_profile_op_name(k // 8).decode(), k % 8): (ms[k], cnt[k]) for k in range(n) if cnt[k]}

    @_on_device
    def adam_step(self, theta, grad, state, lr, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam defaults on the flat meta-parameters (vision/maml_vision.py:85,139-141)."""
        if 'm' not in state:
            state['m'] = torch.zeros_like(theta)
            state['v'] = torch.zeros_like(theta)
            state['step'] = 0
        state['step'] += 1
        _lib.check(self.lib.mi_adam_step(_stream(self.device), _ptr(theta), _ptr(grad), _ptr(state['m']), _ptr(state['v']),
                                         theta.numel(), state['step'], lr, betas[0], betas[1], eps, grad_scale))


def gae_max_rows(state_dim):
    """Longest replay mi_gae_advantages takes for this state dimension (0: unsupported dimension)."""
    return int(_lib.load().mi_gae_max_rows(int(state_dim)))


def gae_advantages(states, next_states, rewards, dones, count, gamma, tau, reg, normalize=True, want_weights=False, weights=None):
    """mi_gae_advantages: cherry-semantics returns -> LinearValue fit -> bootstraps -> GAE (-> ch.normalize) for R replays in one
    launch (reference core_functions/rl.py:95-110,355).  states / next_states [R, B, S], rewards / dones [R, B] fp32 CUDA tensors,
    count [R] int32 (rows in use) or None; weights [R, 2S+4] fp64: use these baseline weights instead of fitting (update_vf=False).
    Returns adv [R, B] fp32 (and the baseline weights [R, 2S+4] fp64)."""
    lib = _lib.load()
    dev = states.device
    R, B, S = states.shape
    f32 = lambda t, shape: t.to(dev, torch.float32).reshape(shape).contiguous()
    states, next_states = f32(states, (R, B, S)), f32(next_states, (R, B, S))
    rewards, dones = f32(rewards, (R, B)), f32(dones, (R, B))
    if count is not None:
        count = count.to(dev, torch.int32).contiguous()
    adv = torch.empty(R, B, dtype=torch.float32, device=dev)
    wts = torch.empty(R, 2 * S + 4, dtype=torch.float64, device=dev) if want_weights else None
    if weights is not None:
        weights = weights.to(dev, torch.float64).reshape(R, 2 * S + 4).contiguous()
    with torch.cuda.device(dev):
        _lib.check(lib.mi_gae_advantages(_stream(dev), _ptr(states), _ptr(next_states), _ptr(rewards), _ptr(dones), _ptr(count), _ptr(weights), R, B, S,
                                         float(gamma), float(tau), float(reg), int(bool(normalize)), _ptr(adv), _ptr(wts)))
    return (adv, wts) if want_weights else adv


def copy_segments(src_ptrs, dst_ptrs, nfloat, npad, dev):
    """mi_copy_segments: segment k = nfloat[k] floats from device address src_ptrs[k] to dst_ptrs[k], zeros up to npad[k] floats (lists of
    ints).  The caller keeps the tensors alive and has checked them (fp32, contiguous, on ``dev``)."""
    n = len(src_ptrs)
    if n == 0:
        return
    vp, u32 = C.c_void_p * n, C.c_uint32 * n
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mi_copy_segments(_stream(dev), vp(*src_ptrs), vp(*dst_ptrs), u32(*nfloat), u32(*npad), n))


def upload_int32(values, dev):
    """A short list of ints as an int32 device tensor through mi_upload_i32: the values travel in kernel arguments (a pageable
    host-to-device copy of a few bytes blocks the host for ~50 us on this stack)."""
    n = len(values)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    if n:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mi_upload_i32(_stream(dev), _ptr(out), (C.c_int32 * n)(*values), n))
    return out


def flatten_parameters(module):
    """Flat fp32 vector in module.parameters() order (what the C ABI calls theta)."""
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()]).float().contiguous()


class PolicyEngine:
    """ctypes wrapper of the MAML-TRPO policy path (mi_policy_* / mi_trpo_*), batched over tasks."""

    def __init__(self, state_size, action_size, hiddens=(100, 100), device=None, activation='relu'):
        self.lib = _lib.load()
        if activation not in ('relu', 'tanh'):
            raise NotImplementedError(f'activation {activation!r}: the reference offers relu and tanh (policies.py:32-37)')
        if not torch.cuda.is_available():
            raise _lib.MiError('PolicyEngine needs a GPU: there is no CPU implementation of the policy path in this package.')
        if len(hiddens) != 2:
            raise ValueError('the HIP policy path implements the reference default: two hidden layers (policies.py:33-34)')
        self.device = _resolve_device(device)
        self.S, self.A, self.H = state_size, action_size, tuple(hiddens)
        desc = _lib.MiPolicyDesc(state_size, action_size, hiddens[0], hiddens[1], int(activation == 'tanh'))
        self._h = C.c_void_p()
        rc = self.lib.mi_policy_create(C.byref(desc), self.device.index, C.byref(self._h))
        if rc:
            raise _lib.MiError(self.lib.mi_policy_last_error(None).decode())
        n = C.c_size_t()
        self.lib.mi_policy_param_count(self._h, C.byref(n))
        self.param_count = n.value
        self._ws = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.mi_policy_destroy(h)
            self._h = C.c_void_p()

    def _check(self, rc):
        if rc:
            raise _lib.MiError(f'libmi_maml policy error {rc}: {self.lib.mi_policy_last_error(self._h).decode()}')

    def _workspace(self, T, B):
        # the general (ANIL-TRPO) plan is a superset of the MAML-TRPO one with the same prefix: one buffer serves both, and a
        # kl_prepare after surrogate finds the surrogate's cached passes where it left them
        b = C.c_size_t()
        self._check(self.lib.mi_trpo_general_workspace_bytes(self._h, T, B, C.byref(b)))
        if self._ws is None or self._ws.numel() < b.value:
            self._ws = torch.empty(b.value, dtype=torch.uint8, device=self.device)
        return self._ws

    @_on_device
    def forward(self, theta, states):
        """loc of the policy density; theta [P] (shared) or [T, P]; states [T, B, S] -> loc [T, B, A]."""
        T, B = states.shape[0], states.shape[1]
        ws = self._workspace(T, B)
        loc = torch.empty(T, B, self.A, device=self.device)
        stride = 0 if theta.dim() == 1 else self.param_count
        self._check(self.lib.mi_policy_forward(self._h, _stream(self.device), _ptr(theta.contiguous()), stride, _ptr(states.contiguous()), T, B,
                                               _ptr(loc), _ptr(ws), ws.numel()))
        return loc

    @_on_device
    def adapt(self, theta, states, actions, adv, count, lr, head_only=False):
        """trpo_update for T tasks: returns (theta_out [T, P], loss [T]).  head_only: ANIL inner loop (body under no_grad)."""
        T, B = states.shape[0], states.shape[1]
        ws = self._workspace(T, B)
        out = torch.empty(T, self.param_count, device=self.device)
        loss = torch.empty(T, device=self.device)
        stride = 0 if theta.dim() == 1 else self.param_count
        self._check(self.lib.mi_policy_adapt(self._h, _stream(self.device), _ptr(theta.contiguous()), stride, _ptr(states), _ptr(actions),
                                             _ptr(adv), _ptr(count), T, B, float(lr), int(head_only), _ptr(out), _ptr(loss), _ptr(ws), ws.numel()))
        return out, loss

    @_on_device
    def surrogate(self, theta, sup, qry, old_loc, old_scale, inner_lr, want_grad):
        """sup/qry: dicts with states [T,B,S], actions [T,B,A], adv [T,B], count [T] int32.  -> (loss, kl, grad or None)."""
        T, B = sup['states'].shape[0], sup['states'].shape[1]
        ws = self._workspace(T, B)
        loss = torch.empty(1, device=self.device)
        kl = torch.empty(1, device=self.device)
        grad = torch.empty(self.param_count, device=self.device) if want_grad else None
        self._check(self.lib.mi_trpo_surrogate(self._h, _stream(self.device), _ptr(theta), _ptr(sup['states']), _ptr(sup['actions']),
                                               _ptr(sup['adv']), _ptr(sup['count']), _ptr(qry['states']), _ptr(qry['actions']),
                                               _ptr(qry['adv']), _ptr(qry['count']), _ptr(old_loc), _ptr(old_scale), T, B,
                                               float(inner_lr), _ptr(loss), _ptr(kl), _ptr(grad), _ptr(ws), ws.numel()))
        return loss, kl, grad

    def _steps_ws(self, T, B, K):
        b = C.c_size_t()
        self._check(self.lib.mi_trpo_steps_workspace_bytes(self._h, T, B, K, C.byref(b)))
        if self._ws is None or self._ws.numel() < b.value:
            self._ws = torch.empty(b.value, dtype=torch.uint8, device=self.device)
        return self._ws

    @_on_device
    def surrogate_steps(self, theta, sup, qry, old_loc, old_scale, inner_lr, want_grad):
        """meta_surrogate_loss with K = sup['states'].shape[0] inner updates (sup arrays carry a leading [K] axis)."""
        K, T, B = sup['states'].shape[0], sup['states'].shape[1], sup['states'].shape[2]
        ws = self._steps_ws(T, B, K)
        loss = torch.empty(1, device=self.device)
        kl = torch.empty(1, device=self.device)
        grad = torch.empty(self.param_count, device=self.device) if want_grad else None
        self._check(self.lib.mi_trpo_surrogate_steps(
            self._h, _stream(self.device), _ptr(theta), K, _ptr(sup['states']), _ptr(sup['actions']), _ptr(sup['adv']), _ptr(sup['count']),
            _ptr(qry['states']), _ptr(qry['actions']), _ptr(qry['adv']), _ptr(qry['count']), _ptr(old_loc), _ptr(old_scale), T, B,
            float(inner_lr), _ptr(loss), _ptr(kl), _ptr(grad), _ptr(ws), ws.numel()))
        return loss, kl, grad

    @_on_device
    def fvp_steps(self, sup, qry, inner_lr, damping, v):
        K, T, B = sup['states'].shape[0], sup['states'].shape[1], sup['states'].shape[2]
        ws = self._steps_ws(T, B, K)
        out = torch.empty(self.param_count, device=self.device)
        self._check(self.lib.mi_trpo_fvp_steps(self._h, _stream(self.device), K, _ptr(sup['states']), _ptr(sup['actions']), _ptr(sup['count']),
                                               _ptr(qry['states']), _ptr(qry['count']), T, B, float(inner_lr), float(damping),
                                               _ptr(v.contiguous()), _ptr(out), _ptr(ws), ws.numel()))
        return out

    @_on_device
    def meta_batch(self, theta, sup, qry, step_batch, inner_lr, loss='a2c', clip=0.1, step_new_old=None, head_only=False,
                   first_order=False, with_grad=True):
        """K = len(step_batch) MAML updates of the policy on replayed support batches + validation loss + meta-gradient for all
        tasks (mi_policy_meta_batch: fast_adapt_vpg / fast_adapt_ppo, reference rl.py:231-255,267-318).
        sup: dict states [NB,T,B,S], actions [NB,T,B,A], adv [NB,T,B], count [NB,T] int32 (NB support batches); qry: the same
        without the NB axis.  Returns (loss [T], theta_adapted [T,P], grad [P] summed over tasks or None)."""
        import numpy as np
        K = len(step_batch)
        T, B = qry['states'].shape[0], qry['states'].shape[1]
        NB = sup['states'].shape[0] if K else 0
        kind = {'a2c': 0, 'ppo': 1, 'dice': 2}[loss]
        sb = (C.c_int32 * max(K, 1))(*[int(x) for x in step_batch])
        if step_new_old is None:
            step_new_old = [1 if (k == 0 or step_batch[k] != step_batch[k - 1]) else 0 for k in range(K)]
        sn = (C.c_int32 * max(K, 1))(*[int(x) for x in step_new_old])
        so = (not first_order) and with_grad
        b = C.c_size_t()
        self._check(self.lib.mi_policy_meta_workspace_bytes(self._h, T, B, K, NB, int(so), C.byref(b)))
        if self._ws is None or self._ws.numel() < b.value:
            self._ws = torch.empty(b.value, dtype=torch.uint8, device=self.device)
        loss_t = torch.empty(T, device=self.device)
        theta_out = torch.empty(T, self.param_count, device=self.device)
        grad = torch.empty(self.param_count, device=self.device) if with_grad else None
        g = lambda d, k: _ptr(d[k].contiguous()) if d is not None and K else C.c_void_p(0)
        if loss == 'dice':       # the DiCE objective couples the samples of an episode: the replays' `dones` travel with them
            if 'done' not in qry or (K and 'done' not in sup):
                raise ValueError("loss='dice' needs the episode-end flags of every replay (key 'done', float32, like 'adv')")
            self._check(self.lib.mi_policy_meta_batch_dones(
                self._h, _stream(self.device), _ptr(theta.contiguous()), K, sb, sn, NB, g(sup, 'states'), g(sup, 'actions'), g(sup, 'adv'),
                g(sup, 'count'), g(sup, 'done'), _ptr(qry['states'].contiguous()), _ptr(qry['actions'].contiguous()),
                _ptr(qry['adv'].contiguous()), _ptr(qry['count'].contiguous()), _ptr(qry['done'].contiguous()), T, B, kind, float(clip),
                float(inner_lr), int(head_only), int(not first_order), int(with_grad), _ptr(loss_t), _ptr(theta_out), _ptr(grad),
                _ptr(self._ws), self._ws.numel()))
            return loss_t, theta_out, grad
        self._check(self.lib.mi_policy_meta_batch(
            self._h, _stream(self.device), _ptr(theta.contiguous()), K, sb, sn, NB, g(sup, 'states'), g(sup, 'actions'), g(sup, 'adv'),
            g(sup, 'count'), _ptr(qry['states'].contiguous()), _ptr(qry['actions'].contiguous()), _ptr(qry['adv'].contiguous()),
            _ptr(qry['count'].contiguous()), T, B, kind, float(clip), float(inner_lr), int(head_only), int(not first_order),
            int(with_grad), _ptr(loss_t), _ptr(theta_out), _ptr(grad), _ptr(self._ws), self._ws.numel()))
        return loss_t, theta_out, grad

    @_on_device
    def kl_prepare(self, theta, sup, qry, old_loc, old_scale, inner_lr, want_grad=False):
        """After ``surrogate`` at the same theta: the context of the exact KL Hessian-vector product for new != old (ANIL-TRPO).
        Returns d mean KL / d theta [P] if ``want_grad``."""
        T, B = sup['states'].shape[0], sup['states'].shape[1]
        ws = self._workspace(T, B)
        grad = torch.empty(self.param_count, device=self.device) if want_grad else None
        self._check(self.lib.mi_trpo_kl_prepare(self._h, _stream(self.device), _ptr(theta), _ptr(sup['states']), _ptr(sup['actions']),
                                                _ptr(sup['count']), _ptr(qry['states']), _ptr(qry['count']), _ptr(old_loc),
                                                _ptr(old_scale), T, B, float(inner_lr), _ptr(grad), _ptr(ws), ws.numel()))
        return grad

    @_on_device
    def fvp_general(self, theta, sup, qry, old_scale, inner_lr, damping, v):
        """Exact Hessian-vector product of the mean KL at the theta of the preceding ``surrogate`` + ``kl_prepare`` calls."""
        T, B = sup['states'].shape[0], sup['states'].shape[1]
        ws = self._workspace(T, B)
        out = torch.empty(self.param_count, device=self.device)
        self._check(self.lib.mi_trpo_fvp_general(self._h, _stream(self.device), _ptr(theta), _ptr(sup['states']), _ptr(sup['actions']),
                                                 _ptr(sup['count']), _ptr(qry['states']), _ptr(qry['count']), _ptr(old_scale), T, B,
                                                 float(inner_lr), float(damping), _ptr(v.contiguous()), _ptr(out), _ptr(ws), ws.numel()))
        return out

    @_on_device
    def fvp(self, theta, sup, qry, inner_lr, damping, v):
        """Fisher-vector product at the theta of the preceding ``surrogate`` call (same batches)."""
        T, B = sup['states'].shape[0], sup['states'].shape[1]
        ws = self._workspace(T, B)
        out = torch.empty(self.param_count, device=self.device)
        self._check(self.lib.mi_trpo_fvp(self._h, _stream(self.device), _ptr(theta), _ptr(sup['states']), _ptr(sup['actions']),
                                         _ptr(sup['count']), _ptr(qry['states']), _ptr(qry['count']), T, B, float(inner_lr),
                                         float(damping), _ptr(v.contiguous()), _ptr(out), _ptr(ws), ws.numel()))
        return out
