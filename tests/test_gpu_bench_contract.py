"""bench.py's output contract on the GPU box: ONE JSON line on stdout with the driver's keys, BASELINE's metric / unit, the
`roofline` and `cpu_baseline` objects, and a `config.workload` naming the BASELINE configuration -- checked on the quickest workload
(cfg1: Omniglot 5-way 1-shot first-order MAML, meta-batch 4), as a child process exactly as the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'cfg1'],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                    # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'tasks/sec' and d['unit'] == 'tasks/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] in ('weak', 'strong')
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config'] and 'Omniglot' in d['config']['workload']
    assert d['value'] > 0 and abs(d['value'] - 4 * 1e3 / d['ms_per_step']) < 1e-2 * d['value']      # 4 tasks per step
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] in ('port', 'reference') and cb['value'] > 0 and cb['cores'] >= 1
    # the engine's post-adaptation figures sit beside the oracle's on the same tasks
    pa = d['post_adapt']
    assert pa['compared_tasks'] >= 1 and pa['max_abs_acc_diff_per_task'] == 0.0 and pa['max_abs_loss_diff_per_task'] < 1e-3
    # what "f32" means on this build is spelled out next to it
    assert d['arithmetic']['split_bf16_operands'] in (True, False) and 'fp32' in d['arithmetic']['note']
    # the shader clock the numbers were taken at (None only where rocm-smi is missing): the roofline peaks assume 2400 MHz
    assert 'clock' in d
    if d['clock'] is not None:
        assert 500 <= d['clock']['sclk_mhz'] <= 2600 and d['clock']['nominal_mhz'] == 2400
    # plain `python bench.py` runs its step through a single-rank RCCL group: the collective record is there at N = 1 too
    co = d['collective']
    assert co is not None and 'error' not in co, co
    assert co['world_size'] == 1 and co['backend'].startswith('nccl') and co['theta_checksum_identical_on_all_ranks'] is True
    assert d['post_adapt']['task_pool'].startswith('8 resident batches')


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell (what the driver's scaling run issues): bench.py launches the two ranks itself and
    relays rank 0's line.  Here the ranks share the box's one card, so the collective goes over gloo (RCCL refuses two ranks on one
    GPU); everything else -- self_launch, torch.distributed.run, init_process_group, the sharded step, the all-reduce, Adam -- is the
    N > 1 path."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['MI_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--workload', 'cfg1',
                        '--pool', '2'], capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['global_meta_batch'] == 8 and d['config']['tasks_per_gpu'] == 4
    assert d['value'] > 0 and abs(d['value'] - 8 * 1e3 / d['ms_per_step']) < 1e-2 * d['value']
    co = d['collective']
    assert co['world_size'] == 2 and co['backend'].startswith('gloo') and co['rccl_version'] is None
    assert co['theta_checksum_identical_on_all_ranks'] is True
    # N > 1 under the default weak scaling: the line also carries the strong-scaling reading (global meta-batch kept, each rank its share)
    st = d['strong_scaling']
    assert st['scaling'] == 'strong' and st['global_meta_batch'] == 4 and st['tasks_per_rank'] == 2 and st['n_gpus'] == 2
    assert st['ms_per_iteration'] > 0 and abs(st['tasks_per_s'] - 4 * 1e3 / st['ms_per_iteration']) < 1e-2 * st['tasks_per_s']
    assert d['secondary']['strong_scaling'] == st
    assert 0.0 <= d['secondary']['valid_acc_mean'] <= 1.0          # (the in-place all-reduce sums the ranks' accuracies: divided by the world size)
