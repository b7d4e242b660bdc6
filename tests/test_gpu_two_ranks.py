"""The N > 1 path with the REAL engine behind it: two ranks share the box's one card (collective over gloo, MI_DIST_BACKEND), each runs
its shard of the meta-batch through its own MetaEngine, one all-reduce, the same Adam step on both.  Checked against the
single-process run over the whole meta-batch: tests/test_sharding_gloo.py covers the same plumbing on the CPU with the oracle as the
compute; here the HIP engine and the collective meet."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _two_ranks(mode, tmp_path, backend='gloo'):
    """Both ranks as child processes (2 + this process = 3 with the card open; the box allows 6).  backend 'gloo': the ranks share card 0;
    'nccl' (= RCCL): rank r on GPU r."""
    port = _free_port()
    procs, outs = [], []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank if backend == 'nccl' else 0), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), MI_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0')
        out = str(tmp_path / f'{mode}_rank{rank}.pt')
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'two_rank_worker.py'), mode, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(log)
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    return [torch.load(o, weights_only=False) for o in outs]


def test_two_ranks_of_the_engine_reduce_to_the_single_process_meta_gradient(tmp_path):
    _check_trainer(*_two_ranks('trainer', tmp_path))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL wants one GPU per rank: runs where the box has two')
def test_two_ranks_over_rccl_take_the_packed_in_place_all_reduce(tmp_path):
    """The same two-rank step with the collective the product uses: backend "nccl" (RCCL over xGMI), rank r on GPU r -- the packed
    [meta-gradient | losses | accuracies] view (sharding.packed_outputs, an as_strided view over the engine's output allocation) all-reduced
    in place by RCCL with two ranks, which the one-GPU boxes of the build rounds cannot exercise (reference vision/maml_vision.py:139-141)."""
    _check_trainer(*_two_ranks('trainer', tmp_path, backend='nccl'))


def _check_trainer(r0, r1):
    sys.path.insert(0, HERE)
    import two_rank_worker as W
    assert r0['world'] == r1['world'] == 2
    assert r0['local_tasks'] == [6, 7, 8] and r1['local_tasks'] == [9, 10, 11]      # second iteration's shards
    # every rank holds the same reduced gradient and the same parameters after two Adam steps, bit for bit (no broadcast needed)
    assert torch.equal(r0['grad'], r1['grad']) and torch.equal(r0['grad2'], r1['grad2']) and torch.equal(r0['theta'], r1['theta'])
    assert r0['loss'] == r1['loss'] and r0['acc'] == r1['acc'] and r0['loss2'] == r1['loss2']
    one = W.trainer_step()                                   # this process, no process group: all six tasks in one engine call
    assert one['world'] == 1 and one['local_tasks'] == list(range(6, 12))
    assert abs(one['loss'] - r0['loss']) <= 1e-6 * max(1.0, abs(one['loss'])) and abs(one['acc'] - r0['acc']) <= 1e-6
    # iteration 1: the same per-task gradients, summed 3 + 3 across ranks instead of 6 in one fold: fp32 summation order only
    assert float((one['grad'] - r0['grad']).abs().max()) <= 2e-6 * float(one['grad'].abs().max())
    # second iteration: the parameters the first Adam step left.  Adam divides by sqrt(v) ~ |g|, so a last-ulp difference of the summed
    # gradient (3 + 3 tasks vs 6 in one fold) can move an element whose gradient is ~0 by a visible fraction of lr: compare what is
    # well conditioned -- the reduced gradient of iteration 2 and the metrics
    scale = float(one['grad2'].abs().max())
    assert float((one['grad2'] - r0['grad2']).abs().max()) <= 2e-4 * scale
    assert abs(one['loss2'] - r0['loss2']) <= 1e-4 * max(1.0, abs(one['loss2'])) and abs(one['acc2'] - r0['acc2']) <= 1e-6
    assert float((one['theta'] - r0['theta']).abs().max()) <= 2 * 0.003 * 2 + 1e-6           # at most lr * (sign flips) per step


def test_two_ranks_of_the_maml_driver_agree_with_one(tmp_path):
    """vision/maml_vision.py under a 2-rank launch: same logged metrics as the single-process run, identical model (parameters and
    BatchNorm buffers -- the running statistics ride on the same all-reduce) on both ranks."""
    sys.path.insert(0, HERE)
    import two_rank_worker as W
    r0, r1 = _two_ranks('driver', tmp_path)
    assert list(r0['sd'].keys()) == list(r1['sd'].keys())
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), k
    assert len(r0['logs']) == W.DRIVER['num_iterations'] and r1['logs'] == []                  # rank 0 logs
    one = W.driver_run(str(tmp_path / 'ckpt_single'))
    for k in ('train_loss', 'train_acc', 'valid_loss', 'valid_acc'):
        assert abs(one['metrics'][k] - r0['metrics'][k]) <= 2e-4 * max(1.0, abs(one['metrics'][k])), (k, one['metrics'], r0['metrics'])
    for k, v in one['sd'].items():
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(r0['sd'][k]), k
        elif 'running_' in k:
            assert float((v - r0['sd'][k]).abs().max()) <= 1e-4 * max(1.0, float(v.abs().max())), k
    assert sorted(os.listdir(tmp_path / 'ckpt_rank0' / 'model_checkpoints')) == ['model_0.pt', 'model_1.pt']
    assert not os.path.exists(tmp_path / 'ckpt_rank1' / 'model_checkpoints')                  # only rank 0 writes


def test_two_ranks_of_the_trpo_driver_keep_one_policy(tmp_path):
    """rl/maml_trpo.py under a 2-rank launch: each rank rolls out and adapts its own tasks (its own sampler stream), the
    Fisher-vector products, the surrogate and its gradient are averaged across ranks inside meta_optimize_trpo -- so CG, the line
    search and the accepted step are the same on both, and the policies stay identical bit for bit without a broadcast."""
    r0, r1 = _two_ranks('trpo', tmp_path)
    assert len(r0['logs']) == 2 and r1['logs'] == []
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), k
        assert torch.isfinite(r0['sd'][k]).all()
