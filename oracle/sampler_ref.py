"""CPU restatement of the task-sampling path (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Reference: utils/data_pre.py:16-112 (learn2learn TaskDataset + task transforms; learn2learn is not vendored and its version
is unpinned => PARITY UNPINNED at this boundary; the structural properties below are what prepare_batch, data_pre.py:115-129,
relies on).  The pixel gather itself is byte/index work and is checked bit-exactly.
"""
import numpy as np


def gather_tasks(dataset, index, rot=None):
    """data[t, r] = dataset[index[t, r]] as float32, rotated counter-clockwise by rot[t, r] quarter turns in the (H, W) plane
    (PIL / torchvision `rotate(angle)` turns counter-clockwise; RandomClassRotation, data_pre.py:34)."""
    index = np.asarray(index)
    out = np.empty(index.shape + dataset.shape[1:], dtype=np.float32)
    for t in range(index.shape[0]):
        for r in range(index.shape[1]):
            img = dataset[index[t, r]].astype(np.float32)
            k = int(rot[t, r]) if rot is not None else 0
            out[t, r] = np.rot90(img, k=k, axes=(1, 2))
    return out


def check_task_structure(index, labels, rot, dataset_labels, ways, shots, classes=None):
    """The invariants of NWays / KShots(2*shots) / RemapLabels / ConsecutiveLabels (/ RandomClassRotation) on one task."""
    k = 2 * shots
    assert index.shape == (ways * k,) and labels.shape == (ways * k,)
    orig = dataset_labels[index]
    groups = orig.reshape(ways, k)
    assert (groups == groups[:, :1]).all(), 'rows of one class must be contiguous (ConsecutiveLabels)'
    cls = groups[:, 0]
    assert len(set(cls.tolist())) == ways, 'NWays: distinct classes'
    assert (np.diff(cls) > 0).all(), 'ConsecutiveLabels: classes in ascending original-label order'
    if classes is not None:
        assert set(cls.tolist()) <= set(int(c) for c in classes), 'FilterLabels'
    assert len(set(index.tolist())) == ways * k, 'KShots without replacement: distinct images'
    new = labels.reshape(ways, k)
    assert (new == new[:, :1]).all() and sorted(new[:, 0].tolist()) == list(range(ways)), 'RemapLabels: a permutation of 0..ways-1'
    if rot is not None:
        rr = rot.reshape(ways, k)
        assert (rr == rr[:, :1]).all(), 'RandomClassRotation: one angle per class'
    # what prepare_batch (data_pre.py:122-127) then produces: `shots` rows of every class in each half, same label order
    assert (labels[0::2] == labels[1::2]).all()
