"""Oracle (TEST INFRASTRUCTURE): autograd restatement of the reference's vision MAML/ANIL path on the CPU.

Functional (no nn.Module cloning); parameters are an ordered ``dict`` keyed with the reference's state_dict names, in the
reference's ``module.parameters()`` registration order.  Every function cites the reference lines it follows.
dtype is whatever the parameter/data tensors carry (fp64 for the oracle proper, fp32 for the "reference-fp32" leg).
"""

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- model specs
def convbase_spec(hidden=64, channels=1, max_pool=False, layers=4, max_pool_factor=1.0):
    """ConvBase ctor arguments (vision_models.py:127-146)."""
    return dict(hidden=hidden, channels=channels, max_pool=max_pool, layers=layers, max_pool_factor=max_pool_factor)


def mini_imagenet_spec(ways, hidden=32, layers=4):
    """MiniImagenetCNN (vision_models.py:93-105): ConvBase(hidden, channels=3, max_pool=True, factor=4//layers) + Linear(25*hidden, ways)."""
    return dict(kind='min', ways=ways, in_shape=(3, 84, 84),
                base=convbase_spec(hidden, 3, True, layers, 4 // layers), fc_in=25 * hidden)


def omniglot_spec(ways, hidden=64, layers=4):
    """OmniglotCNN (vision_models.py:39-49): ConvBase(hidden, channels=1, max_pool=False) + mean(dim=[2,3]) + Linear(hidden, ways)."""
    return dict(kind='omni', ways=ways, in_shape=(1, 28, 28),
                base=convbase_spec(hidden, 1, False, layers, 1.0), fc_in=hidden)


def param_shapes(spec, prefix_base='base.', with_head=True):
    """Parameter names/shapes in registration order (ConvBlock registers normalize before conv, vision_models.py:168-186)."""
    b = spec['base'] if 'base' in spec else spec
    shapes = OrderedDict()
    cin = b['channels']
    for i in range(b['layers']):
        shapes[f'{prefix_base}{i}.normalize.weight'] = (b['hidden'],)
        shapes[f'{prefix_base}{i}.normalize.bias'] = (b['hidden'],)
        shapes[f'{prefix_base}{i}.conv.weight'] = (b['hidden'], cin, 3, 3)
        shapes[f'{prefix_base}{i}.conv.bias'] = (b['hidden'],)
        cin = b['hidden']
    if with_head and 'ways' in spec:
        shapes['linear.weight'] = (spec['ways'], spec['fc_in'])
        shapes['linear.bias'] = (spec['ways'],)
    return shapes


# ----------------------------------------------------------------------------------------------- forward
def conv_block(x, p, i, base, prefix='base.'):
    """ConvBlock.forward (vision_models.py:188-193): conv -> BatchNorm2d (TRAIN mode, batch stats; the reference never
    calls .eval()) -> ReLU -> max_pool / identity.  Stride logic from the ctor (:157-165)."""
    stride = int(2 * base['max_pool_factor'])
    conv_stride = 1 if base['max_pool'] else stride
    x = F.conv2d(x, p[f'{prefix}{i}.conv.weight'], p[f'{prefix}{i}.conv.bias'], stride=conv_stride, padding=1)
    x = F.batch_norm(x, None, None, p[f'{prefix}{i}.normalize.weight'], p[f'{prefix}{i}.normalize.bias'],
                     training=True, momentum=0.1, eps=1e-5)
    x = F.relu(x)
    if base['max_pool']:
        x = F.max_pool2d(x, kernel_size=stride, stride=stride, ceil_mode=False)
    return x


def conv_base(x, p, base, prefix='base.', upto=None):
    """ConvBase (vision_models.py:121-146): Sequential of ConvBlocks."""
    n = base['layers'] if upto is None else upto
    for i in range(n):
        x = conv_block(x, p, i, base, prefix)
    return x


def model_forward(x, p, spec):
    """MiniImagenetCNN.forward (vision_models.py:107-110) / OmniglotCNN.forward (:51-55)."""
    if spec['kind'] == 'omni':
        x = conv_base(x.view(-1, 1, 28, 28), p, spec['base'])
        x = x.mean(dim=[2, 3])
    else:
        x = conv_base(x, p, spec['base'])
        x = x.view(-1, spec['fc_in'])
    return F.linear(x, p['linear.weight'], p['linear.bias'])


def head_forward(x, p):
    """ANIL head: torch.nn.Linear(fc_neurons, ways) (anil_vision.py:93)."""
    return F.linear(x, p['weight'], p['bias'])


# ----------------------------------------------------------------------------------------------- data split
def prepare_batch_indices(n, shots, ways):
    """prepare_batch masks (data_pre.py:122-125): support = positions arange(shots*ways)*2, query = complement."""
    adapt = np.zeros(n, dtype=bool)
    adapt[np.arange(shots * ways) * 2] = True
    return np.nonzero(adapt)[0], np.nonzero(~adapt)[0]


def prepare_batch(data, labels, shots, ways, features=None):
    """prepare_batch (data_pre.py:115-129); ``features`` (ANIL) runs on ALL images before the split (:118-119)."""
    if features is not None:
        data = features(data)
    si, qi = prepare_batch_indices(data.shape[0], shots, ways)
    si, qi = torch.from_numpy(si), torch.from_numpy(qi)
    return data[si], labels[si], data[qi], labels[qi]


def accuracy(predictions, targets):
    """accuracy (vision.py:21-23)."""
    predictions = predictions.argmax(dim=1).view(targets.shape)
    return (predictions == targets).sum().to(predictions.device, torch.float32) / targets.size(0)


# ----------------------------------------------------------------------------------------------- l2l semantics
def clone_params(p):
    """learn2learn ``clone_module``: every parameter becomes ``p.clone()`` (graph edge back to the meta-parameters)."""
    return OrderedDict((k, v.clone()) for k, v in p.items())


def maml_adapt(loss, p, lr, first_order):
    """learn2learn ``MAML.adapt`` + ``maml_update`` (call site vision.py:13; restated in vision/README.md:68-80, and used
    verbatim for RL at rl.py:368-374): g = grad(loss, params, retain_graph=create_graph=second_order); p <- p - lr*g."""
    so = not first_order
    grads = torch.autograd.grad(loss, list(p.values()), retain_graph=so, create_graph=so)
    return OrderedDict((k, v - lr * g) for (k, v), g in zip(p.items(), grads))


def fast_adapt(data, labels, forward, p, adaptation_steps, shots, ways, lr, first_order, features=None):
    """fast_adapt (vision.py:6-18) with CrossEntropyLoss(reduction='mean') (maml_vision.py:86).
    ``forward(x, params)`` is the learner's forward.  Returns (valid_loss, valid_accuracy, adapted params)."""
    ad, al, ed, el = prepare_batch(data, labels, shots, ways, features)
    for _ in range(adaptation_steps):
        train_loss = F.cross_entropy(forward(ad, p), al)
        p = maml_adapt(train_loss, p, lr, first_order)
    predictions = forward(ed, p)
    valid_loss = F.cross_entropy(predictions, el)
    return valid_loss, accuracy(predictions, el), p, predictions


def maml_meta_batch(theta, spec, datas, labelss, adaptation_steps, shots, ways, lr, first_order, backward=True):
    """One meta-iteration's train half (maml_vision.py:102-114): per task clone -> fast_adapt -> eval_loss.backward()
    (grads SUM into the meta-parameters; the caller divides by meta_batch_size, :139-140).
    ``theta``: OrderedDict of leaf tensors.  Returns (losses[T], accs[T], meta_grad OrderedDict, logits list)."""
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in theta.items())
    fwd = lambda x, p: model_forward(x, p, spec)
    losses, accs, logits = [], [], []
    for data, labels in zip(datas, labelss):
        learner = clone_params(leaves)
        with torch.set_grad_enabled(True):
            loss, acc, _, pred = fast_adapt(data, labels, fwd, learner, adaptation_steps, shots, ways, lr, first_order)
            if backward:
                loss.backward()
        losses.append(loss.detach())
        accs.append(acc)
        logits.append(pred.detach())
    grad = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in leaves.items())
    return torch.stack(losses), torch.stack(accs), grad, logits


def anil_meta_batch(theta_feat, theta_head, base, fc_neurons, datas, labelss, adaptation_steps, shots, ways, lr,
                    first_order=False, backward=True):
    """ANIL train half (anil_vision.py:86-94,116-122): features = Sequential(ConvBase, view(-1, fc_neurons)) applied to all
    2*S*W images inside prepare_batch; head = MAML(Linear) (second order by default)."""
    fl = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in theta_feat.items())
    hl = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in theta_head.items())
    feats = lambda x: conv_base(x, fl, base, prefix='0.').view(-1, fc_neurons)
    losses, accs = [], []
    for data, labels in zip(datas, labelss):
        learner = clone_params(hl)
        loss, acc, _, _ = fast_adapt(data, labels, head_forward, learner, adaptation_steps, shots, ways, lr,
                                     first_order, features=feats)
        if backward:
            loss.backward()
        losses.append(loss.detach())
        accs.append(acc)
    gf = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in fl.items())
    gh = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in hl.items())
    return torch.stack(losses), torch.stack(accs), gf, gh


def flatten_params(p):
    """Flat vector in the reference's parameter order (what ``maml.parameters()`` iterates, maml_vision.py:85,139)."""
    return torch.cat([v.reshape(-1) for v in p.values()])


def adam_step(theta, grad, m, v, step, lr=0.003, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (maml_vision.py:85: Adam(maml.parameters(), outer_lr)); flat tensors, in place."""
    step += 1
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    denom = (v.sqrt() / np.sqrt(1 - b2 ** step)).add_(eps)
    theta.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))
    return step
