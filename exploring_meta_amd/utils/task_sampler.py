"""On-device task sampling: learn2learn ``TaskDataset.sample()`` for a dataset that lives in HBM.

Reference: utils/data_pre.py:16-112 builds ``l2l.data.TaskDataset(dataset, task_transforms=[FilterLabels?, NWays(ways),
KShots(2*shots), LoadData, RemapLabels, ConsecutiveLabels, RandomClassRotation?], num_tasks=...)`` and the training loop
calls ``tasks.sample()`` once per task (vision/maml_vision.py:103,116), which loads ``2*shots*ways`` images on the host and
copies them to the device.  Here the whole dataset is resident on the GPU (Mini-ImageNet train split: 0.8 GB as bytes, 3.3 GB
as fp32, of 288 GB), the host draws only the image *indices* of a meta-batch (a few KB) and one HIP launch (mi_sample_tasks)
gathers -- and for Omniglot rotates -- the pixels straight into the ``[T, 2*shots*ways, C, H, W]`` batch the engine consumes.

learn2learn is not vendored by the reference (parity unpinned at this boundary); the transform semantics restated here:
  FilterLabels(labels)      keep the samples of the given classes                      (data_pre.py:29,41,53)
  NWays(n)                  n distinct classes, uniformly
  KShots(k)                 k distinct samples of every chosen class, uniformly (replacement=False)
  RemapLabels(shuffle=True) task labels 0..n-1, assigned to the chosen classes in random order
  ConsecutiveLabels         rows grouped by class, classes in ascending ORIGINAL label order -- prepare_batch's even/odd row
                            split (data_pre.py:122-127) relies on this grouping, not on the label values
  RandomClassRotation(degs) one angle per class and task, applied to all of its images (quarter turns only)
  num_tasks=N               a task is a pure function of its id in [0, N): sampling an id twice yields the same task
"""
import numpy as np
import torch

from .. import _lib


class ResidentDataset:
    """images [N, C, H, W] (uint8 raw pixels or float32) and labels [N], images kept on the GPU; the label -> indices table
    of learn2learn's MetaDataset stays on the host."""

    def __init__(self, images, labels, device='cuda'):
        images = torch.as_tensor(images)
        if images.dim() != 4 or images.dtype not in (torch.uint8, torch.float32):
            raise ValueError('images must be [N, C, H, W] uint8 or float32')
        labels = np.asarray(labels).astype(np.int64).ravel()
        if labels.shape[0] != images.shape[0]:
            raise ValueError('one label per image')
        if (images.shape[1] * images.shape[2] * images.shape[3]) % 4:
            raise ValueError('C*H*W must be a multiple of 4')
        self.images = images.contiguous().to(device)
        self.labels = labels
        self.labels_to_indices = {int(c): np.nonzero(labels == c)[0] for c in np.unique(labels)}

    def __len__(self):
        return self.images.shape[0]


class TaskSampler:
    def __init__(self, dataset, ways, shots, classes=None, rotations=None, remap_shuffle=True, num_tasks=-1, seed=0):
        self.dataset, self.ways, self.shots = dataset, int(ways), int(shots)
        classes = sorted(dataset.labels_to_indices) if classes is None else sorted(int(c) for c in classes)
        missing = [c for c in classes if c not in dataset.labels_to_indices]
        if missing:
            raise ValueError(f'classes not in the dataset: {missing[:5]}')
        if len(classes) < self.ways:
            raise ValueError(f'{len(classes)} classes cannot form {self.ways}-way tasks')
        short = [c for c in classes if len(dataset.labels_to_indices[c]) < 2 * self.shots]
        if short:
            raise ValueError(f'classes with fewer than 2*shots = {2 * self.shots} samples: {short[:5]}')
        self.classes = np.asarray(classes, dtype=np.int64)
        if rotations is not None:
            rot = [float(r) for r in rotations]
            if any(r % 90.0 for r in rot):
                raise ValueError('only multiples of 90 degrees (the reference uses [0, 90, 180, 270], data_pre.py:34)')
            if dataset.images.shape[2] != dataset.images.shape[3]:
                raise ValueError('rotations need square images')
            rotations = np.asarray([int(r // 90) % 4 for r in rot], dtype=np.uint8)
        self.rotations = rotations
        self.remap_shuffle = bool(remap_shuffle)
        self.num_tasks = int(num_tasks)
        self.seed = int(seed)
        self._rng = np.random.default_rng([self.seed, 0x5a3])
        self._lib = None

    # ---------------------------------------------------------------------------------------------------- host side (indices)
    def task_description(self, rng):
        """(image index [n2], task label [n2], quarter turns [n2] or None) of one task drawn from `rng`."""
        k = 2 * self.shots
        chosen = np.sort(rng.choice(self.classes, size=self.ways, replace=False))          # NWays; ConsecutiveLabels order
        new_label = rng.permutation(self.ways) if self.remap_shuffle else np.arange(self.ways)   # RemapLabels
        index = np.concatenate([rng.choice(self.dataset.labels_to_indices[int(c)], size=k, replace=False) for c in chosen])
        labels = np.repeat(new_label.astype(np.int64), k)
        rot = None
        if self.rotations is not None:
            rot = np.repeat(rng.choice(self.rotations, size=self.ways), k).astype(np.uint8)   # one angle per class
        return index.astype(np.int64), labels, rot

    def sample_indices(self, tasks):
        """Indices of `tasks` tasks: (index [T, n2] int64, labels [T, n2] int64, rot [T, n2] uint8 or None), host arrays."""
        out = []
        for _ in range(tasks):
            if self.num_tasks > 0:
                tid = int(self._rng.integers(self.num_tasks))
                out.append(self.task_description(np.random.default_rng([self.seed, 0x7a5c, tid])))
            else:
                out.append(self.task_description(self._rng))
        index = np.stack([o[0] for o in out])
        labels = np.stack([o[1] for o in out])
        rot = np.stack([o[2] for o in out]) if self.rotations is not None else None
        return index, labels, rot

    # -------------------------------------------------------------------------------------------------- device side (pixels)
    def gather(self, index, rot=None):
        """data [T, n2, C, H, W] fp32 on the dataset's device from host index / rotation arrays (mi_sample_tasks)."""
        import ctypes as C
        if self._lib is None:
            self._lib = _lib.load()
        ds = self.dataset.images
        if not ds.is_cuda:
            raise RuntimeError('the task sampler gathers on the GPU (mi_sample_tasks); the dataset must be resident there')
        index = np.ascontiguousarray(index, dtype=np.int64)
        if index.min() < 0 or index.max() >= len(self.dataset):
            raise IndexError('image index out of range')
        T, n2 = index.shape
        idx_d = torch.from_numpy(index).to(ds.device)
        rot_d = torch.from_numpy(np.ascontiguousarray(rot, dtype=np.uint8)).to(ds.device) if rot is not None else None
        out = torch.empty((T, n2) + tuple(ds.shape[1:]), dtype=torch.float32, device=ds.device)
        rc = self._lib.mi_sample_tasks(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(ds.data_ptr()),
                                       int(ds.dtype == torch.uint8), len(self.dataset), ds.shape[1], ds.shape[2], ds.shape[3],
                                       C.c_void_p(idx_d.data_ptr()), C.c_void_p(rot_d.data_ptr() if rot_d is not None else 0),
                                       T, n2, C.c_void_p(out.data_ptr()))
        _lib.check(rc)
        return out

    def sample_batch(self, tasks):
        """A meta-batch: (data [T, 2*shots*ways, C, H, W] fp32, labels [T, 2*shots*ways] int64), both on the GPU."""
        index, labels, rot = self.sample_indices(tasks)
        return self.gather(index, rot), torch.from_numpy(labels).to(self.dataset.images.device)

    def sample(self):
        """One task, the return value of learn2learn's ``tasks.sample()``: (data [n2, C, H, W], labels [n2])."""
        data, labels = self.sample_batch(1)
        return data[0], labels[0]
