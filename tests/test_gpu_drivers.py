"""The driver counterparts of vision/maml_vision.py and rl/maml_trpo.py run end to end on the GPU (short runs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_maml_vision_driver_trains():
    from exploring_meta_amd.vision import maml_vision
    p = dict(maml_vision.params, ways=5, shots=1, adapt_steps=1, meta_batch_size=4, num_iterations=3, inner_lr=0.4)
    logs = []
    model, metrics = maml_vision.run('omni', p, first_order=False, log=logs.append)
    assert len(logs) == 3 and 0.0 <= metrics['test_acc'] <= 1.0
    assert all(torch.isfinite(q).all() for q in model.parameters())


def test_maml_trpo_driver_runs():
    from exploring_meta_amd.rl import maml_trpo
    p = dict(maml_trpo.params, meta_batch_size=3, adapt_batch_size=4, max_path_length=20, num_iterations=2)
    logs = []
    policy = maml_trpo.run(p, log=logs.append)
    assert len(logs) == 2 and all(torch.isfinite(q).all() for q in policy.parameters())


def test_anil_vision_driver_trains():
    from exploring_meta_amd.vision import anil_vision
    p = dict(anil_vision.params, ways=5, shots=1, adapt_steps=2, meta_batch_size=4, num_iterations=3, inner_lr=0.1)
    logs = []
    (features, head), metrics = anil_vision.run('omni', p, log=logs.append)
    assert len(logs) == 3 and 0.0 <= metrics['valid_acc'] <= 1.0
    before = torch.nn.Linear(128, 5).weight                       # the head and the trunk both moved away from their init
    assert all(torch.isfinite(q).all() for q in list(features.parameters()) + list(head.parameters()))
    assert head.module.weight.shape == before.shape
