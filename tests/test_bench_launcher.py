"""bench.py starts its own ranks: `python bench.py --gpus N` from a bare shell (no torchrun environment) must run N ranks under
torch.distributed.run, relay rank 0's JSON line on stdout and exit with the job's code -- the driver's SCALE run is exactly that
command.  Covered here on the CPU with the `--launch-check` leg (ranks join the process group over gloo, all-reduce a one, rank 0
prints): the launcher branch, the rendezvous on 127.0.0.1 and the relay are the real ones, only the engine is left out.
Reference: the collective of a data-parallel run sits at vision/maml_vision.py:139-141."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _bare_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(MI_DIST_BACKEND='gloo', **extra)
    return env


def test_self_launch_starts_two_ranks_and_relays_rank0s_line():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--launch-check'], capture_output=True, text=True, timeout=300, cwd=REPO,
                       env=_bare_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                                  # one rank prints
    d = json.loads(lines[0])
    assert d == {'launch_check': True, 'world_size': 2, 'allreduce_of_ones': 2.0, 'backend': 'gloo'}


def test_self_launch_relays_a_failing_job():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--launch-check', '--workload', 'no-such-workload'], capture_output=True,
                       text=True, timeout=300, cwd=REPO, env=_bare_env())
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_self_launch_only_from_a_bare_shell(monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    for k in ('WORLD_SIZE', 'RANK'):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch(['--gpus', '1', '--steps', '3']) is None           # N = 1: in-process
    assert bench.self_launch(['--steps', '3']) is None                          # default N = 1
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    assert bench.self_launch(['--gpus', '2']) is None                           # already one of torchrun's ranks
