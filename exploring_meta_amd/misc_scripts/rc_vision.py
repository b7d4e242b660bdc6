#!/usr/bin/env python3
"""Representation-change experiment (reference misc_scripts/rc_vision.py:34-99,150-165): adapt a clone of the model to each
task and collect, per layer, the representation of the adaptation data before and after adaptation.  The similarity measures
(CCA / CKA, ``utils/cca.py``, ``utils/cka.py``) run on the returned arrays."""
import numpy as np

from ..core_functions import accuracy, prepare_batch

default_params = {"adapt_steps": 1, "inner_lr": 0.1, "n_tasks": 5, "layers": [0, 1, 2, 3, 4]}


def get_rep_from_batch(model, batch, layer=4):
    """reference rc_vision.py:150-165: [features, batch] matrix of the layer's representation (layer -1: the logits)."""
    if layer == -1:
        return model(batch).cpu().detach().numpy()
    rep = model.get_rep_i(batch, layer).cpu().detach().numpy()
    b, c, h, w = rep.shape
    return rep.reshape((c * h * w, b))


def run_rep_exp(model, loss, tasks, device, ways, shots, rep_params=default_params):
    """Returns (acc_results [n_tasks, 2] = (adapted, initial) accuracy on the evaluation half, reps) with
    reps[layer] = list over tasks of (adapted_rep, init_rep) as produced by get_rep_from_batch."""
    init_model = model.clone()
    adapt_model = model.clone()                     # adapted cumulatively over the tasks, like the reference (:48-70)
    acc = np.zeros((rep_params['n_tasks'], 2))
    reps = {int(layer): [] for layer in rep_params['layers']}
    for t in range(rep_params['n_tasks']):
        adapt_d, adapt_l, eval_d, eval_l = prepare_batch(tasks.sample(), shots, ways, device)
        for _ in range(rep_params['adapt_steps']):
            train_error = loss(adapt_model(adapt_d), adapt_l)
            train_error = train_error / len(adapt_d)                     # reference :69
            adapt_model.adapt(train_error)
            acc[t, 0] = accuracy(adapt_model(eval_d), eval_l).item()
            acc[t, 1] = accuracy(init_model(eval_d), eval_l).item()
        for layer in reps:
            reps[layer].append((get_rep_from_batch(adapt_model, adapt_d, layer), get_rep_from_batch(init_model, adapt_d, layer)))
    return acc, reps
