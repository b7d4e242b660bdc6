#!/usr/bin/env python3
"""Benchmark of the MAML hot path on MI355X: tasks/sec for BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one meta-iteration of the train half of the reference loop (vision/maml_vision.py:93-141): every rank
processes its shard of the meta-batch (K inner steps on support, query forward, second-order outer backward) through
mi_meta_batch_maml, the flat meta-gradient is all-reduced over RCCL, and the identical Adam step is applied on every rank.
Inputs (synthetic tasks) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# The engine forks weight gradients onto a side stream and RCCL brings its own: with the HIP default of 4 hardware queues the
# streams of one process collide on a queue and the overlap turns into a 5 % loss (measured: 26.5 vs 25.2 ms per step under
# torch.distributed); 8 queues keep every stream on its own.  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.utils import synthetic  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (fits one GPU)
    'cfg2': dict(name='Mini-ImageNet 5-way 5-shot second-order MAML, 4-conv-32, 5 adapt steps, meta-batch 32 per GPU',
                 dataset='min', ways=5, shots=5, steps=5, lr=0.5, first_order=False, tasks=32),
    # BASELINE.json configs[2]: ANIL, 64-filter trunk once on all 50 rows, head-only inner loop (anil_vision.py defaults: K=1)
    'cfg3': dict(name='Mini-ImageNet 5-way 5-shot ANIL (head-only inner loop), 4-conv-64 trunk, 1 adapt step, meta-batch 32 per GPU',
                 dataset='min', ways=5, shots=5, steps=1, lr=0.5, first_order=False, tasks=32, anil=True),
    'cfg4': dict(name='Mini-ImageNet 5-way 1-shot second-order MAML, 1 adapt step, 32 tasks per GPU',
                 dataset='min', ways=5, shots=1, steps=1, lr=0.5, first_order=False, tasks=32),
    'cfg1': dict(name='Omniglot 5-way 1-shot first-order MAML, meta-batch 4', dataset='omni', ways=5, shots=1, steps=1,
                 lr=0.5, first_order=True, tasks=4),
}

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak

# MiniImagenetCNN-32 conv geometry (input hw, ci) per block; co = 32 everywhere
MIN_LAYERS = [(84, 3), (42, 32), (21, 32), (10, 32)]


def conv_launch_flops(layer, n_img):
    """Algorithmic FLOPs of one conv launch over n_img images (2*9*ci*co per output pixel; SURVEY.md 8d per-image figures)."""
    hw, ci = MIN_LAYERS[layer]
    return 2.0 * 9 * ci * 32 * hw * hw * n_img


def conv_launch_bytes(layer, n_img):
    """Algorithmic HBM bytes of one conv+stats launch: read the layer input once, write the conv output once."""
    hw, ci = MIN_LAYERS[layer]
    return 4.0 * hw * hw * (ci + 32) * n_img


def init_theta(spec, seed=42):
    """The reference's initialisers (xavier-uniform weights, zero biases, BatchNorm gamma ~ U(0,1): vision_models.py:175,204-207)
    drawn from the build's hash generator."""
    shapes = dict(spec.param_shapes())
    w = synthetic.ref_init_weights(shapes, seed)
    return torch.from_numpy(np.concatenate([w[k].ravel() for k in shapes])).float()


def cpu_baseline(wl, budget_s=20.0, max_tasks=16):
    """The oracle (CPU restatement of the reference loop, fp32, all host cores) on a bounded sample of the same workload."""
    from collections import OrderedDict
    from oracle import vision_ref as R
    spec = R.mini_imagenet_spec(wl['ways']) if wl['dataset'] == 'min' else R.omniglot_spec(wl['ways'])
    host_cores = os.cpu_count() or 1
    w = synthetic.hash_weights(R.param_shapes(spec), 42)
    theta = OrderedDict((k, torch.from_numpy(v).float()) for k, v in w.items())

    anil = None
    if wl.get('anil'):
        base = R.convbase_spec(hidden=64, channels=3, max_pool=True)
        shapes = R.param_shapes(base, prefix_base='0.', with_head=False)
        tf = OrderedDict((k, torch.from_numpy(v).float()) for k, v in synthetic.hash_weights(shapes, 42).items())
        th = OrderedDict((k, torch.from_numpy(v).float()) for k, v in
                         synthetic.hash_weights(OrderedDict([('weight', (wl['ways'], 1600)), ('bias', (wl['ways'],))]), 43).items())
        anil = (tf, th, base)

    def run(task_ids):
        datas, labels = [], []
        for t in task_ids:
            d, l = synthetic.make_task(wl['dataset'], t, wl['ways'], wl['shots'])
            datas.append(torch.from_numpy(d))
            labels.append(torch.from_numpy(l))
        t0 = time.perf_counter()
        if anil:
            R.anil_meta_batch(anil[0], anil[1], anil[2], 1600, datas, labels, wl['steps'], wl['shots'], wl['ways'], wl['lr'],
                              wl['first_order'])
        else:
            R.maml_meta_batch(theta, spec, datas, labels, wl['steps'], wl['shots'], wl['ways'], wl['lr'], wl['first_order'])
        return time.perf_counter() - t0

    # PyTorch-CPU autograd on 5..25-image batches does not scale to hundreds of threads: pick the fastest intra-op thread
    # count on a small probe (one first-order single-step task) and use that for the baseline.
    probe = dict(wl, steps=1, first_order=True)

    def probe_time(nthreads):
        torch.set_num_threads(nthreads)
        d, l = synthetic.make_task(probe['dataset'], 0, probe['ways'], probe['shots'])
        args = (theta, spec, [torch.from_numpy(d)], [torch.from_numpy(l)], 1, probe['shots'], probe['ways'], probe['lr'], True)
        R.maml_meta_batch(*args)
        t0 = time.perf_counter()
        R.maml_meta_batch(*args)
        return time.perf_counter() - t0

    cands = sorted({c for c in (4, 8, 16, 32, 64) if c <= host_cores} | {min(host_cores, 8)})
    cores = min(cands, key=probe_time)
    torch.set_num_threads(cores)
    run([0])                                   # warm-up (thread pools, oneDNN primitives)
    done, elapsed = 0, 0.0
    while done < max_tasks and elapsed < budget_s:
        elapsed += run([done])
        done += 1
    return dict(value=done / elapsed, unit='tasks/s', cores=cores, kind='port',
                sample=f'{done} tasks of the same workload, sequential per-task loop, PyTorch-CPU autograd fp32 '
                       f'(oracle/vision_ref.py), {cores} intra-op threads (fastest of {cands} on a probe; host has '
                       f'{host_cores} logical cores)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--tasks', type=int, default=0, help='override the tasks per GPU of the workload (sweeps; not a BASELINE configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', default='', help='write a per-kernel event-time breakdown (one extra untimed step) to this file')
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.tasks:
        wl['tasks'] = args.tasks
        wl['name'] += f' [tasks per GPU overridden: {args.tasks}]'

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run always go through RCCL (also exercised at N=1)
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    T = wl['tasks']
    if wl.get('anil'):
        spec = ModelSpec.anil(wl['ways'])
    else:
        spec = ModelSpec.mini_imagenet(wl['ways']) if wl['dataset'] == 'min' else ModelSpec.omniglot(wl['ways'])
    eng = MetaEngine(spec)
    run_batch = eng.meta_batch_anil if wl.get('anil') else eng.meta_batch
    theta = init_theta(spec).cuda()
    task_ids = [rank * T + i for i in range(T)]          # shard by global task id: rank r owns tasks [rT, (r+1)T)
    data, labels = synthetic.make_meta_batch(wl['dataset'], task_ids, wl['ways'], wl['shots'])
    data = torch.from_numpy(data).cuda()
    labels = torch.from_numpy(labels).cuda()
    adam = {}
    out = {}
    from exploring_meta_amd.sharding import MetaTrainer

    def compute(th, _task_ids):                 # this rank's shard is already resident in HBM (data, labels)
        loss, acc, grad, _ = run_batch(th, data, labels, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
        return loss, acc, grad

    def adam_fn(th, grad, scale):               # maml_vision.py:139-141
        eng.adam_step(th, grad, adam, 0.003, grad_scale=scale)

    trainer = MetaTrainer(compute, adam_fn, T * world)     # one flat all-reduce (RCCL) of the meta-gradient per iteration

    def step():
        out['loss'], out['acc'], _ = trainer.step(theta)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    # Dominant kernel = the conv-family (op, block) with the largest share of one meta-iteration's HIP-event time (blocks 2-4:
    # the kernels with a clean algorithmic FLOP count).  Found on one untimed, fully profiled step; the timed region then
    # records events around exactly that kernel's launches.
    n_img = T * wl['ways'] * wl['shots']
    CONV_OPS = {'conv_fwd_stats': (1, 'conv3x3_mfma_kernel<32,1,EPI_STATS,fwd>'), 'dgrad': (1, 'conv3x3_mfma_kernel<32,1,EPI_NONE,dgrad>'),
                'wgrad': (1, 'wgrad3x3_rows_mfma_kernel (1 term)'), 'tangent_conv_fwd': (2, 'conv3x3_mfma_kernel<32,2,EPI_TSTATS,fwd>'),
                'tangent_dgrad': (2, 'conv3x3_mfma_kernel<32,2,EPI_NONE,dgrad>'), 'tangent_wgrad': (2, 'wgrad3x3_rows_mfma_kernel (2 terms)')}
    dom = None
    if args.workload == 'cfg2':
        eng.set_overlap(False)                  # additive per-kernel times for the selection step
        eng.profile(True)
        step()
        torch.cuda.synchronize()
        first = eng.profile_collect()
        eng.profile(False)
        eng.set_overlap(True)
        # the weight-gradient kernels run on the engine's side stream concurrently with other work: their in-run durations are
        # not standalone figures, so the roofline kernel is chosen among the main-stream convs
        cands = {k: v for k, v in first.items() if k[0] in CONV_OPS and k[1] >= 1 and 'wgrad' not in k[0]}
        dom = max(cands, key=lambda k: cands[k][0])
        eng.profile(True, *dom)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_collect() if dom else {}
    eng.profile(False)
    if dist is not None:
        tmax = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    # Secondary figure: the reference runs one validation fast_adapt per train task without backward
    # (maml_vision.py:117-124); here that half is one more fused call with with_grad=0 on T other tasks.
    vdata, vlabels = synthetic.make_meta_batch(wl['dataset'], [10_000 + t for t in task_ids], wl['ways'], wl['shots'])
    vdata, vlabels = torch.from_numpy(vdata).cuda(), torch.from_numpy(vlabels).cuda()

    def valid():
        return run_batch(theta, vdata, vlabels, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'], with_grad=False)

    valid()
    nsec = max(2, min(5, args.steps))
    fence()
    t1 = time.perf_counter()
    for _ in range(nsec):
        step()
        vloss, vacc, _, _ = valid()
    fence()
    dt_tv = (time.perf_counter() - t1) / nsec
    if dist is not None:
        tmax = torch.tensor([dt_tv], device='cuda', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_tv = tmax.item()
    secondary = {'metric': 'iterations/sec (train + validation halves)', 'value': round(1.0 / dt_tv, 3), 'ms_per_iteration': round(dt_tv * 1e3, 3),
                 'tasks_per_iteration': f'{T * world} train + {T * world} validation', 'steps': nsec,
                 'valid_acc_mean': round(float(vacc.mean()), 5)}

    # Measured HBM roofline: the build's own streaming-copy kernel, same process, same run (read + write bytes / time).
    import ctypes as C
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device='cuda').zero_()
    dst = torch.empty_like(src)
    cp = lambda: eng.lib.mi_stream_copy(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(src.data_ptr()),
                                        C.c_void_p(dst.data_ptr()), nbytes)
    cp()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cp()
    e1.record()
    torch.cuda.synchronize()
    hbm_copy_gbps = 2.0 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst

    roofline = None
    if dom in prof:
        ms, cnt = prof[dom]
        terms, kname = CONV_OPS[dom[0]]
        hw = MIN_LAYERS[dom[1]][0]
        flops = terms * conv_launch_flops(dom[1], n_img)
        achieved = flops / (ms / cnt * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(REPO, 'profiles', 'pmc_traffic.json')       # HBM bytes/launch from rocprofv3 --pmc passes (profiles/README.md)
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f'{dom[0]},{dom[1]}')
        roofline = dict(kernel=f'{kname}, block {dom[1] + 1} ({hw}x{hw}, 32->32 filters)', op=dom[0], bound='mfma',
                        achieved=round(achieved, 2), peak=FP32_MFMA_PEAK_TF, unit='TFLOP/s',
                        frac=round(achieved / FP32_MFMA_PEAK_TF, 4), traffic=traffic, launches=int(cnt),
                        avg_launch_ms=round(ms / cnt, 4), flops_per_launch=flops,
                        algorithmic_bytes_per_launch=terms * conv_launch_bytes(dom[1], n_img),
                        hbm_stream_copy_measured_GBps=round(hbm_copy_gbps, 1))

    if args.breakdown:                          # every rank runs the extra step (it contains the all-reduce); rank 0 writes
        eng.set_overlap(False)                  # one stream: the per-kernel times add up to the iteration
        eng.profile(True)
        step()
        torch.cuda.synchronize()
        full = eng.profile_collect()
        eng.profile(False)
        eng.set_overlap(True)
        tot = sum(v[0] for v in full.values())
    if args.breakdown and rank == 0:
        with open(args.breakdown, 'w') as f:
            f.write(f'# per-kernel HIP-event time of ONE meta-iteration, workload {args.workload} (T={T} tasks), total {tot:.3f} ms\n')
            f.write('op,layer,launches,total_ms,avg_ms,share\n')
            for (op, layer), (ms, cnt) in sorted(full.items(), key=lambda kv: -kv[1][0]):
                f.write(f'{op},{layer},{cnt},{ms:.4f},{ms / cnt:.4f},{ms / tot:.4f}\n')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl)

    if rank == 0:
        value = world * T * args.steps / dt
        line = {
            'metric': 'tasks/sec', 'value': round(value, 2), 'unit': 'tasks/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': wl['name'], 'tasks_per_gpu': T, 'global_meta_batch': T * world, 'ways': wl['ways'],
                       'shots': wl['shots'], 'adapt_steps': wl['steps'], 'inner_lr': wl['lr'],
                       'second_order': not wl['first_order'], 'parallelism': f'task-sharded dp{world}, 1 all-reduce/iter'},
            'post_adapt': {'query_loss_mean': round(float(out['loss']), 5), 'query_acc_mean': round(float(out['acc']), 5)},
            'secondary': secondary, 'hbm_stream_copy_GBps': round(hbm_copy_gbps, 1), 'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
