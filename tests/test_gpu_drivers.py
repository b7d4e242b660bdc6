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


def test_maml_vision_driver_writes_reference_checkpoints(tmp_path, golden_small):
    """--save_dir: model_checkpoints/model_<it>.pt every save_every iterations and model.pt at the end, with the reference's
    state_dict keys (utils/experiment.py:85-90, maml_vision.py:143-144; keys pinned by golden G6)."""
    from exploring_meta_amd.vision import maml_vision
    p = dict(maml_vision.params, ways=5, shots=1, adapt_steps=1, meta_batch_size=2, num_iterations=3, save_every=2,
             save_dir=str(tmp_path))
    model, _ = maml_vision.run('omni', p, first_order=True, log=lambda *_: None)
    import os
    assert sorted(os.listdir(tmp_path / 'model_checkpoints')) == ['model_0.pt', 'model_2.pt']
    sd = torch.load(tmp_path / 'model.pt')
    assert list(sd.keys()) == list(model.state_dict().keys())
    assert all(torch.equal(sd[k].cpu(), v.cpu()) for k, v in model.state_dict().items())
    fresh = maml_vision.OmniglotCNN(5)
    fresh.load_state_dict(sd)                                 # round trip through the reference key layout
    # BatchNorm buffers as the reference's loop leaves them: every forward pass counted (3 iterations x 2 tasks x train+valid x
    # (adapt_steps + 1) passes), running statistics moved off their initial 0 / 1 (tests/test_gpu_running_stats.py checks the values)
    for i in range(4):
        assert int(sd[f'base.{i}.normalize.num_batches_tracked']) == 3 * 2 * 2 * 2
        assert float(sd[f'base.{i}.normalize.running_mean'].abs().max()) > 0 and float((sd[f'base.{i}.normalize.running_var'] - 1).abs().max()) > 0


def test_anil_vision_driver_writes_reference_checkpoints(tmp_path):
    """ANIL --save_dir: the reference calls save_model_checkpoint(features, 'features_<it+1>') -> model_checkpoints/
    model_features_<it+1>.pt / model_head_<it+1>.pt (utils/experiment.py:89-90, anil_vision.py:151-153) and, after the loop,
    save_model(features, 'features') / save_model(head, 'head') -> features.pt / head.pt (anil_vision.py:163-164)."""
    import os
    from exploring_meta_amd.vision import anil_vision
    p = dict(anil_vision.params, ways=5, shots=1, adapt_steps=1, meta_batch_size=2, num_iterations=3, save_every=2, inner_lr=0.1,
             save_dir=str(tmp_path))
    (features, head), _ = anil_vision.run('omni', p, log=lambda *_: None)
    assert sorted(os.listdir(tmp_path / 'model_checkpoints')) == ['model_features_1.pt', 'model_features_3.pt', 'model_head_1.pt',
                                                                  'model_head_3.pt']
    for name, module in (('features.pt', features), ('head.pt', head)):
        sd = torch.load(tmp_path / name)
        assert list(sd.keys()) == list(module.state_dict().keys())
        assert all(torch.equal(sd[k].cpu(), v.cpu()) for k, v in module.state_dict().items())


def test_maml_trpo_driver_runs():
    from exploring_meta_amd.rl import maml_trpo
    p = dict(maml_trpo.params, meta_batch_size=3, adapt_batch_size=4, max_path_length=20, num_iterations=2)
    logs = []
    policy = maml_trpo.run(p, log=logs.append)
    assert len(logs) == 2 and all(torch.isfinite(q).all() for q in policy.parameters())


def test_anil_trpo_driver_runs():
    """rl/anil_trpo.py counterpart: DiagNormalPolicyANIL, head-only inner loop, meta_optimize_trpo(anil=True) on the exact KL Hessian."""
    from exploring_meta_amd.rl import anil_trpo
    p = dict(anil_trpo.params, meta_batch_size=3, adapt_batch_size=4, max_path_length=20, num_iterations=2)
    logs = []
    policy = anil_trpo.run(p, log=logs.append)
    assert len(logs) == 2 and all(torch.isfinite(q).all() for q in policy.parameters())
    assert type(policy).__name__ == 'DiagNormalPolicyANIL'


def test_anil_vision_driver_trains():
    from exploring_meta_amd.vision import anil_vision
    p = dict(anil_vision.params, ways=5, shots=1, adapt_steps=2, meta_batch_size=4, num_iterations=3, inner_lr=0.1)
    logs = []
    (features, head), metrics = anil_vision.run('omni', p, log=logs.append)
    assert len(logs) == 3 and 0.0 <= metrics['valid_acc'] <= 1.0
    before = torch.nn.Linear(128, 5).weight                       # the head and the trunk both moved away from their init
    assert all(torch.isfinite(q).all() for q in list(features.parameters()) + list(head.parameters()))
    assert head.module.weight.shape == before.shape


@pytest.mark.parametrize('anil', [False, True])
def test_maml_ppo_driver_runs(anil):
    from exploring_meta_amd.rl import maml_ppo
    p = dict(maml_ppo.params, meta_batch_size=3, adapt_batch_size=4, max_path_length=15, num_iterations=2, adapt_steps=2)
    logs = []
    policy = maml_ppo.run(p, anil=anil, log=logs.append)
    assert len(logs) == 2 and all(torch.isfinite(q).all() for q in policy.parameters())
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in policy.parameters())


def test_fast_adapt_vpg_accumulates_meta_gradient():
    from exploring_meta_amd import core_functions as cf
    pol = cf.MAML(cf.DiagNormalPolicy(2, 2).cuda(), lr=0.05)
    gen = torch.Generator(device='cuda').manual_seed(0)
    P = dict(inner_lr=0.05, gamma=0.99, tau=1.0, adapt_steps=2, adapt_batch_size=4)
    total = 0.0
    for goal in ([0.2, -0.1], [-0.3, 0.4]):
        task = cf.Particles2DRunner(goal, 12, gen)
        loss, rew, _ = cf.fast_adapt_vpg(task, pol.clone(), cf.LinearValue(2, 2), P)
        total = total + loss
        assert rew < 0
    (total / 2).backward()
    g = torch.cat([q.grad.reshape(-1) for q in pol.parameters()])
    assert torch.isfinite(g).all() and g.abs().sum() > 0
    with torch.no_grad():                                   # evaluation: no graph, plain tensor
        loss, _, _ = cf.fast_adapt_vpg(cf.Particles2DRunner([0.1, 0.1], 12, gen), pol.clone(), cf.LinearValue(2, 2), P)
    assert not loss.requires_grad


@pytest.mark.parametrize('algo', ['vpg', 'ppo', 'trpo'])
def test_evaluate_rl(algo):
    """reference rl.py:142-196 / evaluate_vpg / evaluate_ppo / evaluate_trpo on Particles2D goals: adaptation must not touch the
    meta-policy, rewards are per-episode sums of negative distances."""
    from exploring_meta_amd import core_functions as cf
    torch.manual_seed(0)
    pol = cf.DiagNormalPolicy(2, 2, activation='tanh').cuda()
    before = pol.flat().clone()
    gen = torch.Generator(device='cuda').manual_seed(1)
    P = dict(inner_lr=0.05, gamma=0.99, tau=1.0, adapt_steps=2, adapt_batch_size=4, max_path_length=12, ppo_epochs=2, ppo_clip_ratio=0.1, n_tasks=3)
    fn = dict(vpg=cf.evaluate_vpg, ppo=cf.evaluate_ppo, trpo=cf.evaluate_trpo)[algo]
    policy = cf.MAML(pol, lr=P['inner_lr']) if algo != 'trpo' else pol
    rewards, mean_rew, mean_suc = fn([[0.3, 0.1], [-0.2, 0.4], [0.0, -0.5]], policy, cf.LinearValue(2, 2), P, generator=gen)
    assert len(rewards) == 3 and all(r < 0 for r in rewards) and mean_rew == pytest.approx(sum(rewards) / 3) and mean_suc == 0.0
    assert torch.equal(pol.flat(), before)


def test_cl_and_rc_experiments_run():
    """misc_scripts counterparts on the step-wise learner: shapes, ranges, and that adaptation changes later layers' reps."""
    import numpy as np
    from exploring_meta_amd import core_functions as cf
    from exploring_meta_amd.misc_scripts import cl_vision, rc_vision
    from exploring_meta_amd.vision.maml_vision import SyntheticTasks
    torch.manual_seed(0)
    model = cf.OmniglotCNN(5).cuda()
    maml = cf.MAML(model, lr=0.1)
    loss = torch.nn.CrossEntropyLoss(reduction='mean')
    dev = torch.device('cuda')
    acc = cl_vision.run_cl_exp(maml, loss, SyntheticTasks('omni', 5, 1, 0), dev, 5, 1, dict(adapt_steps=2, inner_lr=0.1, n_tasks=3))
    assert acc.shape == (3, 3) and (acc >= 0).all() and (acc <= 1).all()
    accs, reps = rc_vision.run_rep_exp(maml, loss, SyntheticTasks('omni', 5, 1, 50), dev, 5, 1,
                                       dict(adapt_steps=1, inner_lr=0.1, n_tasks=2, layers=[0, 1, 4]))
    assert accs.shape == (2, 2) and set(reps) == {0, 1, 4} and len(reps[4]) == 2
    a0, i0 = reps[0][0]
    assert np.array_equal(a0, i0)                                  # layer 0 = the input itself
    a4, i4 = reps[4][-1]
    assert a4.shape == i4.shape == (64 * 2 * 2, 5) and not np.allclose(a4, i4)
