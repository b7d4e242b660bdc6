"""Host mirror of the reference's ``utils/data_pre.py::prepare_batch`` (reference lines 115-129).

The dataset samplers of that file (``get_omniglot`` / ``get_mini_imagenet``, learn2learn TaskDataset pipelines that download
datasets) are out of scope; ``exploring_meta_amd.utils.synthetic`` supplies task batches with
the same layout.  Inside the fused engine the split runs as a HIP kernel (``mi_prepare_batch``); this function is the
stand-alone equivalent for callers that want the four tensors.
"""
import ctypes as C

import numpy as np
import torch


def prepare_batch(batch, shots, ways, device, features=None):
    """Same signature and return value as the reference: (adaptation_data, adaptation_labels, evaluation_data,
    evaluation_labels); support = rows {0,2,...,2(shots*ways-1)}, query = the remaining rows, order preserved."""
    data, labels = batch
    data, labels = data.to(device), labels.to(device)
    if features is not None:
        data = features(data)
    n = data.size(0)
    if n != 2 * shots * ways:
        raise ValueError(f'task batch holds {n} rows, expected 2*shots*ways = {2 * shots * ways}')
    if data.is_cuda and features is None and data.dim() == 4 and data.dtype == torch.float32:
        return _prepare_batch_hip(data.contiguous(), labels.contiguous())
    adaptation_indices = np.zeros(n, dtype=bool)
    adaptation_indices[np.arange(shots * ways) * 2] = True
    evaluation_indices = torch.from_numpy(~adaptation_indices)
    adaptation_indices = torch.from_numpy(adaptation_indices)
    return (data[adaptation_indices], labels[adaptation_indices], data[evaluation_indices], labels[evaluation_indices])


def _prepare_batch_hip(data, labels):
    """The split through the C ABI (mi_prepare_batch writes NHWC); returned as NCHW views like the reference's tensors."""
    from .. import _lib
    lib = _lib.load()
    n2, c, h, w = data.shape
    xs = torch.empty(n2 // 2, h, w, c, device=data.device)
    xq = torch.empty_like(xs)
    ys = torch.empty(n2 // 2, dtype=torch.int32, device=data.device)
    yq = torch.empty_like(ys)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mi_prepare_batch(st, p(data), p(labels.to(torch.int64)), 1, n2, c, h, w, p(xs), p(xq), p(ys), p(yq)))
    return xs.permute(0, 3, 1, 2), ys.long(), xq.permute(0, 3, 1, 2), yq.long()
