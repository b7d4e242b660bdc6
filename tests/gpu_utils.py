"""Helpers for the -m gpu parity tests: device buffers through the C ABI, and a parity report merged back from the GPU box."""
import ctypes as C
import json
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl')


def report(name, **metrics):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    clean = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in metrics.items()}
    with open(REPORT, 'a') as f:
        f.write(json.dumps(dict(test=name, **clean)) + '\n')
    print('[parity]', name, clean)


def dev(a, dtype=torch.float32):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    return t.to(dtype).cuda().contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def max_err(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
