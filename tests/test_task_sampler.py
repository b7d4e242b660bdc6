"""Task sampler: host-side index drawing (CPU tests) and the HIP gather (GPU tests) against oracle/sampler_ref.py.
Reference path: utils/data_pre.py:16-112 + tasks.sample() (vision/maml_vision.py:103,116)."""
import numpy as np
import pytest
import torch

from exploring_meta_amd.utils import synthetic
from exploring_meta_amd.utils.task_sampler import ResidentDataset, TaskSampler
from oracle import sampler_ref as S


def _dataset(n_classes, per_class, c, hw, dtype, device):
    n = n_classes * per_class
    u = synthetic.hash_uniform(91, (n, c, hw, hw))
    imgs = (u * 256).astype(np.uint8) if dtype == 'u8' else u.astype(np.float32)
    labels = np.repeat(np.arange(n_classes) * 3 + 7, per_class)        # non-contiguous original labels
    perm = np.argsort(synthetic.hash_uniform(5, (n,)))                   # interleave the classes
    return imgs[perm], labels[perm], ResidentDataset(torch.from_numpy(imgs[perm]), labels[perm], device=device)


@pytest.mark.parametrize('ways,shots,rotations,shuffle', [(5, 1, None, True), (5, 5, None, False), (20, 1, [0.0, 90.0, 180.0, 270.0], True)])
def test_task_structure(ways, shots, rotations, shuffle):
    imgs, labels, ds = _dataset(30, 12, 1, 8, 'f32', 'cpu')
    classes = sorted(set(labels.tolist()))[:25]
    sm = TaskSampler(ds, ways, shots, classes=classes, rotations=rotations, remap_shuffle=shuffle, seed=3)
    index, lab, rot = sm.sample_indices(16)
    assert index.shape == (16, 2 * shots * ways) and lab.shape == index.shape and (rot is None) == (rotations is None)
    for t in range(16):
        S.check_task_structure(index[t], lab[t], None if rot is None else rot[t], labels, ways, shots, classes)
        if not shuffle:
            assert (lab[t] == np.repeat(np.arange(ways), 2 * shots)).all()
    assert len({tuple(r) for r in index.tolist()}) > 1                   # tasks differ
    again = TaskSampler(ds, ways, shots, classes=classes, rotations=rotations, remap_shuffle=shuffle, seed=3).sample_indices(16)
    assert (again[0] == index).all() and (again[1] == lab).all()          # same seed, same stream of tasks


def test_num_tasks_makes_tasks_a_function_of_their_id():
    imgs, labels, ds = _dataset(10, 6, 1, 8, 'f32', 'cpu')
    sm = TaskSampler(ds, 5, 1, num_tasks=3, seed=1)
    index, lab, _ = sm.sample_indices(40)
    assert len({tuple(r) for r in index.tolist()}) == 3                   # only 3 distinct tasks, each repeated identically
    assert len({tuple(a) + tuple(b) for a, b in zip(index.tolist(), lab.tolist())}) == 3


def test_argument_errors():
    imgs, labels, ds = _dataset(6, 4, 1, 8, 'f32', 'cpu')
    with pytest.raises(ValueError):
        TaskSampler(ds, 7, 1)                                              # more ways than classes
    with pytest.raises(ValueError):
        TaskSampler(ds, 5, 3)                                              # 2*shots > samples per class
    with pytest.raises(ValueError):
        TaskSampler(ds, 5, 1, rotations=[45.0])
    with pytest.raises(ValueError):
        TaskSampler(ds, 5, 1, classes=[1, 2, 3, 4, 5])                     # labels that do not exist
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                                  # the gather has no CPU path
            TaskSampler(ds, 5, 1).sample()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,c,hw,rotations', [('u8', 3, 84, None), ('f32', 1, 28, [0.0, 90.0, 180.0, 270.0]), ('f32', 3, 12, None),
                                                  ('u8', 1, 28, [90.0, 270.0])])
def test_gather_bit_exact(dtype, c, hw, rotations):
    imgs, labels, ds = _dataset(12, 8, c, hw, dtype, 'cuda')
    sm = TaskSampler(ds, 5, 2, rotations=rotations, seed=9)
    index, lab, rot = sm.sample_indices(6)
    got = sm.gather(index, rot).cpu().numpy()
    want = S.gather_tasks(imgs, index, rot)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)
    if rotations is not None:
        assert len(set(rot.ravel().tolist())) > 1
    with pytest.raises(IndexError):
        sm.gather(np.full((1, 20), len(ds), dtype=np.int64))


@pytest.mark.gpu
def test_sampled_batch_feeds_the_engine():
    """tasks.sample() -> fast_adapt, the reference loop (maml_vision.py:103-112), with every pixel staying on the device."""
    from exploring_meta_amd import core_functions as cf
    ways, shots = 5, 1
    n_cls, per = 8, 4
    protos = synthetic.hash_uniform(3, (n_cls, 1, 28, 28))
    noise = synthetic.hash_uniform(4, (n_cls, per, 1, 28, 28))
    imgs = ((protos[:, None] > 0.5) ^ (noise > 0.9)).astype(np.float32).reshape(n_cls * per, 1, 28, 28)
    labels = np.repeat(np.arange(n_cls), per)
    ds = ResidentDataset(torch.from_numpy(imgs), labels)
    sm = TaskSampler(ds, ways, shots, rotations=[0.0, 90.0, 180.0, 270.0], seed=2)
    torch.manual_seed(0)
    model = cf.OmniglotCNN(ways).cuda()
    maml = cf.MAML(model, lr=0.5, first_order=False)
    loss = torch.nn.CrossEntropyLoss()
    batch = sm.sample()
    assert batch[0].shape == (2 * shots * ways, 1, 28, 28) and batch[0].is_cuda and batch[1].dtype == torch.int64
    eval_loss, eval_acc = cf.fast_adapt(batch, maml.clone(), loss, 1, shots, ways, torch.device('cuda'))
    eval_loss.backward()
    assert torch.isfinite(eval_loss) and 0.0 <= eval_acc.item() <= 1.0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    data, lab = sm.sample_batch(4)                                       # the batched entry on a sampled meta-batch
    total, losses, accs = cf.meta_batch_adapt(maml.clone(), data, lab, 1, shots, ways)
    assert losses.shape == (4,) and torch.isfinite(losses).all()
