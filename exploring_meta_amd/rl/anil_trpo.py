#!/usr/bin/env python3
"""Counterpart of the reference's ``rl/anil_trpo.py`` (:104-129) on Particles2D: ``DiagNormalPolicyANIL`` (tanh body, linear head),
``fast_adapt_trpo(..., anil=True, first_order=True)`` with the body under no_grad during the inner updates, and
``meta_optimize_trpo(..., anil=True)`` -- whose KL Hessian-vector product is exact for new != old (mi_trpo_fvp_general).

    python -m exploring_meta_amd.rl.anil_trpo --meta_batch_size 20 --num_iterations 5
"""
import argparse

from .maml_trpo import params as _maml_params, run as _run

# rl/anil_trpo.py:20-41 (inner_lr 0.01, outer_lr 0.1, fc_neurons 100); path length / batch sizes as the Particles2D defaults
params = dict(_maml_params, inner_lr=0.01, outer_lr=0.1, fc_neurons=100)


def run(p, log=print):
    return _run(p, log=log, anil=True)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='ANIL-TRPO on Particles2D (MI355X engine)')
    for k, v in params.items():
        parser.add_argument(f'--{k}', type=type(v), default=v)
    args = parser.parse_args()
    for k in params:
        params[k] = getattr(args, k)
    run(params)
