"""Shared test helpers (oracle-side parameter construction from the hash generator)."""
from collections import OrderedDict

import numpy as np
import torch

from exploring_meta_amd.utils import synthetic
from oracle import vision_ref as R


def hash_params(shapes, seed, dtype=torch.float64):
    w = synthetic.hash_weights(shapes, seed)
    return OrderedDict((k, torch.from_numpy(v).to(dtype)) for k, v in w.items())


def model_params(spec, seed, dtype=torch.float64):
    return hash_params(R.param_shapes(spec), seed, dtype)


def task_tensors(dataset, task_ids, ways, shots, dtype=torch.float64, seed=42):
    datas, labels = [], []
    for t in task_ids:
        d, l = synthetic.make_task(dataset, t, ways, shots, seed)
        datas.append(torch.from_numpy(d).to(dtype))
        labels.append(torch.from_numpy(l))
    return datas, labels


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
