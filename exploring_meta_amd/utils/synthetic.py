"""Seeded synthetic few-shot tasks with the tensor layout the reference's task samplers hand to ``fast_adapt``.

There is no dataset (and no network) on the build/GPU boxes, so benchmarks and parity tests run on
synthetic tasks that have exactly the layout ``utils/data_pre.py:70-112`` (reference) produces:
``data [2*shots*ways, C, H, W]`` float32 and ``labels [2*shots*ways]`` int64 sorted by class
(``[0]*2S + [1]*2S + ...``), so that the even/odd support/query split of ``prepare_batch``
(reference ``utils/data_pre.py:122-127``) yields ``shots`` support and ``shots`` query images per class.

The generator is a counter-based hash (splitmix64 finaliser) written in integer numpy arithmetic, so the
streams are identical on every numpy version/platform: fixtures under ``tests/golden`` store only seeds.
"""

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def hash_uniform(seed, shape, stream=0):
    """float64 uniforms in [0,1), a pure function of (seed, stream, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        base = _mix(np.array([np.uint64(seed) * _GOLDEN + np.uint64(stream) * _M1 + np.uint64(0x1234567)],
                             dtype=np.uint64))
        idx = np.arange(n, dtype=np.uint64)
        z = _mix(base + idx * _GOLDEN)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def hash_normalish(seed, shape, stream=0):
    """Unit-variance, zero-mean Irwin-Hall(4) variate (exact float64 arithmetic, no libm)."""
    u = hash_uniform(seed, (4,) + tuple(shape), stream)
    return (u.sum(axis=0) - 2.0) * np.sqrt(3.0)


def task_labels(ways, shots):
    """Labels as l2l's ``ConsecutiveLabels`` + ``KShots(2*shots)`` arrange them (reference data_pre.py:81-84)."""
    return np.repeat(np.arange(ways, dtype=np.int64), 2 * shots)


def mini_imagenet_task(task_id, ways=5, shots=5, seed=42, hw=84, channels=3, noise=32.0, contrast=1.0):
    """Raw 0..255 float images: class prototype (8x8 block-constant) + sigma=`noise` pixel noise, clipped.  `contrast` < 1 pulls
    the prototypes towards mid-grey (harder tasks: the defaults are separable enough for 100 % post-adaptation accuracy)."""
    s = seed + 1000003 * int(task_id)
    nb = (hw + 7) // 8
    proto = hash_uniform(s, (ways, channels, nb, nb), stream=1) * 255.0
    proto = np.repeat(np.repeat(proto, 8, axis=2), 8, axis=3)[:, :, :hw, :hw]
    if contrast != 1.0:
        proto = 127.5 + contrast * (proto - 127.5)
    n = 2 * shots * ways
    labels = task_labels(ways, shots)
    nz = hash_normalish(s, (n, channels, hw, hw), stream=2)
    data = np.clip(proto[labels] + noise * nz, 0.0, 255.0).astype(np.float32)
    return data, labels


def omniglot_task(task_id, ways=5, shots=1, seed=42, hw=28, flip=0.04):
    """Binary stroke-like images in {0,1} (reference feeds 1-img, data_pre.py:19-21): 7x7 prototype
    upsampled x4 with a fraction `flip` (default 4 %) of the pixels flipped."""
    s = seed + 1000003 * int(task_id)
    cells = (hw + 3) // 4
    proto = (hash_uniform(s, (ways, 1, cells, cells), stream=3) < 0.25).astype(np.float64)
    proto = np.repeat(np.repeat(proto, 4, axis=2), 4, axis=3)[:, :, :hw, :hw]
    n = 2 * shots * ways
    labels = task_labels(ways, shots)
    flipped = hash_uniform(s, (n, 1, hw, hw), stream=4) < flip
    img = proto[labels]
    data = np.where(flipped, 1.0 - img, img).astype(np.float32)
    return data, labels


def make_task(dataset, task_id, ways, shots, seed=42, **hardness):
    if dataset in ('min', 'mini_imagenet'):
        return mini_imagenet_task(task_id, ways, shots, seed, **{k: v for k, v in hardness.items() if k in ('noise', 'contrast')})
    if dataset in ('omni', 'omniglot'):
        return omniglot_task(task_id, ways, shots, seed, **{k: v for k, v in hardness.items() if k == 'flip'})
    raise ValueError(f'Dataset not supported: {dataset}')


def make_meta_batch(dataset, task_ids, ways, shots, seed=42, **hardness):
    """Stack tasks: data [T, 2*S*W, C, H, W] float32, labels [T, 2*S*W] int64."""
    ds, ls = zip(*(make_task(dataset, t, ways, shots, seed, **hardness) for t in task_ids))
    return np.stack(ds), np.stack(ls)


def hash_weights(shapes, seed):
    """Deterministic parameter values for cross-version fixtures.

    ``shapes``: ordered mapping name -> shape using the reference's state_dict key names. Distributions follow the
    reference initialisers (vision_models.py:175,204-207,48-49): xavier-uniform conv/linear weights, zero biases,
    BN gamma ~ U(0,1), BN beta = 0. (Biases/betas get small non-zero values instead so gradients w.r.t. them are
    exercised by the fixtures.)
    """
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        u = hash_uniform(seed, shape, stream=100 + i)
        if name.endswith('normalize.weight'):
            v = 0.1 + 0.9 * u
        elif name.endswith('bias'):
            v = (u - 0.5) * 0.2
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            fan_out = shape[0] * shape[2] * shape[3]
            bound = np.sqrt(6.0 / (fan_in + fan_out))
            v = (2.0 * u - 1.0) * bound
        elif len(shape) == 2:
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = (2.0 * u - 1.0) * bound
        else:
            v = u - 0.5
        out[name] = v.astype(np.float64)
    return out


def ref_init_weights(shapes, seed):
    """The reference's initialisers drawn from the hash generator (vision_models.py:175,204-207,48-49): xavier-uniform conv /
    linear weights, ZERO conv / linear / BatchNorm biases, BatchNorm gamma ~ U(0,1); a 2-d weight named ``linear.weight`` whose
    fan-in is below 100 (OmniglotCNN's Linear(64, ways)) is N(0,1) as in vision_models.py:48."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        u = hash_uniform(seed, shape, stream=100 + i)
        if name.endswith('normalize.weight'):
            v = u
        elif name.endswith('bias'):
            v = np.zeros(shape)
        elif len(shape) == 4:
            bound = np.sqrt(6.0 / ((shape[1] + shape[0]) * shape[2] * shape[3]))
            v = (2.0 * u - 1.0) * bound
        elif len(shape) == 2 and name == 'linear.weight' and shape[1] < 100:
            v = hash_normalish(seed, shape, stream=100 + i)
        elif len(shape) == 2:
            v = (2.0 * u - 1.0) * np.sqrt(6.0 / (shape[0] + shape[1]))
        else:
            v = u - 0.5
        out[name] = v.astype(np.float64)
    return out


def uniform_task(dataset, task_id, ways, shots, seed=42):
    """Plateau-free task: i.i.d. uniform pixels (0..255 for Mini-ImageNet shapes, 0..1 for Omniglot shapes), the layout of
    ``make_task``.  No two pixels are equal, so max-pool / ReLU decisions have no exact ties: the well-conditioned inputs the
    tight parity bars are stated on (SURVEY.md 8c calibrated the one-step tolerances on random inputs)."""
    n = 2 * shots * ways
    s = seed + 7000003 * int(task_id)
    if dataset in ('min', 'mini_imagenet'):
        data = hash_uniform(s, (n, 3, 84, 84), stream=7) * 255.0
    elif dataset in ('omni', 'omniglot'):
        data = hash_uniform(s, (n, 1, 28, 28), stream=7)
    else:
        raise ValueError(f'Dataset not supported: {dataset}')
    return data.astype(np.float32), task_labels(ways, shots)


def make_uniform_meta_batch(dataset, task_ids, ways, shots, seed=42):
    ds, ls = zip(*(uniform_task(dataset, t, ways, shots, seed) for t in task_ids))
    return np.stack(ds), np.stack(ls)
