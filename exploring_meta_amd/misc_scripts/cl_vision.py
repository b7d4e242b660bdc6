#!/usr/bin/env python3
"""Continual-learning accuracy matrix (reference misc_scripts/cl_vision.py:24-81) on the HIP engine's step-wise learner.

For every task i: ``learner = maml.clone()``, ``adapt_steps`` x {``loss(learner(adapt_data_i))``, ``learner.adapt``}, then the
accuracy of that learner on every task j's evaluation data -> ``acc_matrix[i, j]``.  Result files, plots and the CL metrics
(``utils/cl_metrics.py``) are the caller's business."""
import numpy as np

from ..core_functions import accuracy, prepare_batch

default_params = {"adapt_steps": 1, "inner_lr": 0.1, "n_tasks": 10}


def run_cl_exp(maml, loss, tasks, device, ways, shots, cl_params=default_params, features=None, setting=2):
    """``tasks.sample()`` -> (data, labels) like a learn2learn TaskDataset.  setting 1 evaluates on the adaptation data itself,
    otherwise on the held-out half (reference :35-45).  Returns acc_matrix [n_tasks, n_tasks] (rows: adapted on, columns:
    evaluated on)."""
    pool = []
    for _ in range(cl_params['n_tasks']):
        adapt_d, adapt_l, eval_d, eval_l = prepare_batch(tasks.sample(), shots, ways, device, features=features)
        pool.append({'adapt': (adapt_d, adapt_l), 'eval': (adapt_d, adapt_l) if setting == 1 else (eval_d, eval_l)})
    acc_matrix = np.zeros((cl_params['n_tasks'], cl_params['n_tasks']))
    for i, task_i in enumerate(pool):
        adapt_d, adapt_l = task_i['adapt']
        learner = maml.clone()
        for _ in range(cl_params['adapt_steps']):
            learner.adapt(loss(learner(adapt_d), adapt_l))
        for j, task_j in enumerate(pool):
            eval_d, eval_l = task_j['eval']
            acc_matrix[i, j] = accuracy(learner(eval_d), eval_l).item()
    return acc_matrix
