#!/usr/bin/env python3
"""Benchmark of the MAML hot path on MI355X: tasks/sec for BASELINE.json's configurations (headline: configs[1]).

    python bench.py --gpus N --steps K --warmup W [--workload cfg2] [--scaling weak|strong]
        N > 1 from a bare shell: bench.py starts its own ranks (python -m torch.distributed.run --nproc-per-node N, before anything
        touches the GPU) and relays rank 0's line and the job's exit code; under torch.distributed.run it is one of the ranks.
        N = 1: a single-rank RCCL process group in-process, so the step's all-reduce and the `collective` record are the N > 1 code path.

A "step" is one meta-iteration of the train half of the reference loop (vision/maml_vision.py:93-141): every rank processes its
shard of the meta-batch (K inner steps on support, query forward, second-order outer backward) through mi_meta_batch_maml, the
flat meta-gradient is all-reduced over RCCL, and the identical Adam step is applied on every rank.  For cfg5 a step is one
meta_optimize_trpo on resident replays (rl/maml_trpo.py:130-134).  Inputs (synthetic tasks) are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.

--scaling weak (default): every rank runs the workload's tasks-per-GPU (32), global meta-batch 32*N -- BASELINE config 4's
reading (256 tasks sharded 32/GPU over 8 GPUs).  --scaling strong: the global meta-batch stays at the workload's size (32),
rank r runs its contiguous share of 32/N tasks.
"""
import argparse
import json
import os
import sys
import time


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as a CHILD job (never an exec; this process
    has not touched the GPU -- nothing but the standard library is imported yet) and exit with its code.  The ranks' stdout is this
    process's stdout, so rank 0's JSON line arrives unchanged; torchrun's own chatter goes to stderr.
    Reference: the all-reduce point of a data-parallel run is vision/maml_vision.py:139-141."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--gpus', type=int, default=1)
    known, _ = ap.parse_known_args(argv)
    if known.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return None
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this pool: RCCL needs it in every rank
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={known.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


_RESULT_OUT = None
if __name__ == '__main__':
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)
    # stdout carries ONE line, the result.  Libraries write there too (RCCL prints a five-line version banner with printf when its
    # first communicator comes up): keep the real stdout for the result and point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    _RESULT_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

# The engine forks weight gradients onto a side stream and RCCL brings its own: with the HIP default of 4 hardware queues the
# streams of one process collide on a queue and the overlap turns into a 5 % loss (measured: 26.5 vs 25.2 ms per step under
# torch.distributed); 8 queues keep every stream on its own.  Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import ctypes as C

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from exploring_meta_amd.engine import MetaEngine, ModelSpec  # noqa: E402
from exploring_meta_amd.sharding import MetaTrainer, init_process_group, shard_range  # noqa: E402
from exploring_meta_amd.utils import roofline as RF  # noqa: E402
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
    # BASELINE.json configs[4]: rl/maml_trpo.py on Particles2D, 20 tasks x 20 episodes x 100 steps, 1 inner step
    'cfg5': dict(name='Particles2D MAML-TRPO, 2x100 MLP policy, 20 tasks per meta-batch, 20 episodes x 100 steps per replay, '
                      '1 inner adapt step', kind='trpo', tasks=20),
}
# Synthetic-task hardness used by the benchmark: weaker class prototypes and stronger pixel noise than the generator's defaults,
# so that the post-adaptation accuracy of cfg2 at the initial parameters sits near 0.74 instead of saturating (measured sweep in
# profiles/r2/hardness_sweep.txt; the one-step configurations are at chance level at initialisation whatever the inputs -- one
# step at lr 0.5 from random weights overshoots, loss 12..17 -- so their comparison with the oracle is per task, not a mean).
HARDNESS = {'min': dict(contrast=0.3, noise=64.0), 'omni': dict(flip=0.04)}

HBM_PEAK_GBS = RF.PEAK_GBPS
FP32_MFMA_PEAK_TF = RF.PEAK_TFLOPS


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def init_theta(spec, seed=42):
    """The reference's initialisers (xavier-uniform weights, zero biases, BatchNorm gamma ~ U(0,1): vision_models.py:175,204-207)
    drawn from the build's hash generator."""
    shapes = dict(spec.param_shapes())
    w = synthetic.ref_init_weights(shapes, seed)
    return torch.from_numpy(np.concatenate([w[k].ravel() for k in shapes])).float()


def make_batch(wl, task_ids):
    """Synthetic task batches [T, 2*S*W, C, H, W] / labels, generated task by task on a small thread pool (the hash generator is numpy
    integer arithmetic, which releases the GIL: a 32-task Mini-ImageNet batch takes seconds on one core)."""
    from concurrent.futures import ThreadPoolExecutor
    task_ids = list(task_ids)
    one = lambda t: synthetic.make_task(wl['dataset'], t, wl['ways'], wl['shots'], **HARDNESS[wl['dataset']])
    with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1, len(task_ids)))) as ex:
        ds, ls = zip(*ex.map(one, task_ids))
    return np.stack(ds), np.stack(ls)


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_vision(wl, spec, theta_flat, budget_s=20.0, max_tasks=16):      # max_tasks = n_cmp of run_vision
    """The oracle (CPU restatement of the reference loop, fp32) on a bounded sample of the same workload: the first tasks of
    rank 0's shard from the same initial parameters.  Returns (baseline dict, per-task oracle loss, accuracy)."""
    from collections import OrderedDict
    from oracle import vision_ref as R
    host_cores = os.cpu_count() or 1
    named, off = OrderedDict(), 0
    for k, shp in spec.param_shapes():
        n = int(np.prod(shp))
        named[k] = theta_flat[off:off + n].reshape(shp).clone()
        off += n
    if wl.get('anil'):
        rspec = None
        base = R.convbase_spec(hidden=64, channels=3, max_pool=True)
        tf = OrderedDict(('0.' + k[len('base.'):], v) for k, v in named.items() if k.startswith('base.'))
        th = OrderedDict([('weight', named['linear.weight']), ('bias', named['linear.bias'])])
    else:
        rspec = R.mini_imagenet_spec(wl['ways']) if wl['dataset'] == 'min' else R.omniglot_spec(wl['ways'])

    def run(task_ids, steps=None, fo=None):
        datas, labels = [], []
        for t in task_ids:
            d, l = synthetic.make_task(wl['dataset'], t, wl['ways'], wl['shots'], **HARDNESS[wl['dataset']])
            datas.append(torch.from_numpy(d))
            labels.append(torch.from_numpy(l))
        K = wl['steps'] if steps is None else steps
        f = wl['first_order'] if fo is None else fo
        t0 = time.perf_counter()
        if wl.get('anil'):
            out = R.anil_meta_batch(tf, th, base, 1600, datas, labels, K, wl['shots'], wl['ways'], wl['lr'], f)
        else:
            out = R.maml_meta_batch(named, rspec, datas, labels, K, wl['shots'], wl['ways'], wl['lr'], f)
        return time.perf_counter() - t0, out[0], out[1]

    # PyTorch-CPU autograd on 5..25-image batches does not scale to hundreds of threads: pick the fastest intra-op thread
    # count on a small probe (one first-order single-step task) and use that for the baseline.
    def probe_time(nthreads):
        torch.set_num_threads(nthreads)
        run([0], steps=1, fo=True)
        return run([0], steps=1, fo=True)[0]

    cands = sorted({c for c in (4, 8, 16, 32, 64) if c <= host_cores} | {min(host_cores, 8)})
    cores = min(cands, key=probe_time)
    torch.set_num_threads(cores)
    run([0])                                   # warm-up (thread pools, oneDNN primitives)
    done, elapsed, losses, accs = 0, 0.0, [], []
    while done < max_tasks and elapsed < budget_s:
        dt, l, a = run([done])
        elapsed += dt
        losses.append(float(l[0]))
        accs.append(float(a[0]))
        done += 1
    base_line = dict(value=done / elapsed, unit='tasks/s', cores=cores, kind='port', cpu=cpu_model(), host_logical_cores=host_cores,
                     sample=f'{done} tasks of the same workload (task ids 0..{done - 1}, same initial parameters), sequential per-task '
                            f'loop, PyTorch-CPU autograd fp32 (oracle/vision_ref.py), {cores} intra-op threads (fastest of {cands} on a probe)')
    return base_line, losses, accs


def pick_dominant(spec, prof, n_img):
    """Among the ops with an algorithmic work count (conv family, BatchNorm streams, fused block-1 kernels) on the MAIN stream, the
    (op, layer) with the largest HIP-event time in one fully profiled step."""
    cands = {}
    for (op, layer), (ms, cnt) in prof.items():
        if ('wgrad' in op or 'dgrad' in op) and layer >= 1:
            # weight gradients of blocks >= 2 run on the engine's side stream, and the dgrad of the same block is launched beside
            # them on the main stream: two matrix-bound kernels share the chip, so their launch durations in the timed region
            # (and in a rocprofv3 trace) are about twice their isolated ones and say nothing about the kernel.  Their isolated
            # figures are in `roofline.single_stream_step` below.
            continue
        if layer >= spec.n_layers or RF.op_costs(spec, op, layer, n_img) is None:
            continue
        cands[(op, layer)] = ms
    return max(cands, key=cands.get) if cands else None


def run_vision(args, wl, rank, world, local, dist):
    Tw = wl['tasks']
    if args.scaling == 'strong':
        lo, hi = shard_range(Tw, rank, world)
        task_ids, global_T = list(range(lo, hi)), Tw
        if hi - lo < 1:
            raise SystemExit(f'--scaling strong: {Tw} tasks cannot be shared by {world} ranks')
    else:
        task_ids, global_T = [rank * Tw + i for i in range(Tw)], Tw * world       # rank r owns tasks [r*T, (r+1)*T)
    T = len(task_ids)
    if wl.get('anil'):
        spec = ModelSpec.anil(wl['ways'])
    else:
        spec = ModelSpec.mini_imagenet(wl['ways']) if wl['dataset'] == 'min' else ModelSpec.omniglot(wl['ways'])
    eng = MetaEngine(spec)
    use_graph = wl.get('graph', False) if args.graph < 0 else bool(args.graph)
    eng.set_overlap(not args.no_overlap)
    eng.set_graph(use_graph)          # theta is updated in place and the task batches stay resident: every step repeats the same call
    run_batch = eng.meta_batch_anil if wl.get('anil') else eng.meta_batch
    theta0 = init_theta(spec)
    theta = theta0.cuda()
    # A resident POOL of distinct task batches, rotated through by the step loop: every step adapts to tasks the parameters have not
    # just been trained on, as the reference's loop does (tasks.sample() per task, maml_vision.py:103), so the activation statistics --
    # and with them the power draw at the socket cap -- stay those of training instead of those of one memorised batch.  Batch b of
    # rank r holds the global task ids b * (tasks per iteration) + (this rank's ids): every world size sees the same task set per step.
    per_iter = global_T
    pool = []
    for b in range(max(1, args.pool)):
        d, l = make_batch(wl, [b * per_iter + t for t in task_ids])
        pool.append((torch.from_numpy(d).cuda(), torch.from_numpy(l).cuda()))
    data, labels = pool[0]
    adam, out = {}, {'n': 0}

    def compute(th, _task_ids):                 # this rank's shards are already resident in HBM (pool)
        d, l = pool[out['n'] % len(pool)]
        out['n'] += 1
        loss, acc, grad, _ = run_batch(th, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
        return loss, acc, grad

    def adam_fn(th, grad, scale):               # maml_vision.py:139-141
        eng.adam_step(th, grad, adam, 0.003, grad_scale=scale)

    trainer = MetaTrainer(compute, adam_fn, global_T)      # one flat all-reduce (RCCL) of the meta-gradient per iteration

    def step():
        out['loss'], out['acc'], _ = trainer.step(theta)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # post-adaptation loss / accuracy at the INITIAL parameters on the tasks the CPU oracle will also run (task ids 0..15; untimed)
    n_cmp = 16
    cdata, clabels = make_batch(wl, list(range(n_cmp)))
    l0, a0, _, _ = run_batch(theta, torch.from_numpy(cdata).cuda(), torch.from_numpy(clabels).cuda(), wl['shots'], wl['steps'], wl['lr'],
                             first_order=wl['first_order'], with_grad=False)
    eng_loss0, eng_acc0 = l0.cpu().numpy().astype(np.float64), a0.cpu().numpy().astype(np.float64)
    del cdata, clabels
    lall, aall, _, _ = run_batch(theta, data, labels, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'], with_grad=False)
    init_loss_mean, init_acc_mean = float(lall.mean()), float(aall.mean())

    for _ in range(args.warmup):
        step()

    # Dominant kernel: found on one untimed, fully profiled step with everything on one stream (additive per-kernel times); the
    # timed region then records HIP events around exactly that kernel's launches, on the stream they are launched on.
    n_img = T * wl['ways'] * wl['shots'] * (2 if wl.get('anil') else 1)
    eng.set_overlap(False)
    eng.profile(True)
    step()
    torch.cuda.synchronize()
    first = eng.profile_collect()
    eng.profile(False)
    eng.set_overlap(not args.no_overlap)
    dom = pick_dominant(spec, first, n_img)
    if dom:
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
    acc_after_timed, loss_after_timed = float(out['acc']), float(out['loss'])

    secondary = None
    if not args.no_secondary:
        # Secondary figure: the reference runs one validation fast_adapt per train task without backward
        # (maml_vision.py:117-124).  MAML: ONE fused call over [train tasks | validation tasks] (mi_meta_batch_maml_tv) -- the validation
        # tasks' K support steps and query forward ride in the train tasks' launches, only the backward half is train-only.  ANIL: a second
        # call with with_grad = 0.
        vdata, vlabels = make_batch(wl, [10_000_000 + t for t in task_ids])
        vdata, vlabels = torch.from_numpy(vdata).cuda(), torch.from_numpy(vlabels).cuda()
        from exploring_meta_amd.sharding import packed_outputs
        if wl.get('anil'):
            def step_tv():
                step()
                vl, va, _, _ = run_batch(theta, vdata, vlabels, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'], with_grad=False)
                return va
            how = 'train call + a second call with with_grad = 0'
        else:
            tv = [(torch.cat([d, vdata]), torch.cat([l, vlabels])) for d, l in pool[:2]]

            def step_tv():
                d, l = tv[out['n'] % len(tv)]
                out['n'] += 1
                loss, acc, grad, _ = eng.meta_batch(theta, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'], grad_tasks=T)
                flat = packed_outputs(grad, loss, acc)          # [meta-gradient | losses of both halves | accuracies of both halves]
                if dist is not None:
                    # (the in-place SUM makes slot i of the accuracies the sum over ranks of their i-th task's accuracy: the mean below
                    # divides by the world size; without the packed view only the gradient is reduced, like MetaTrainer.step's fallback)
                    if flat is not None:
                        dist.all_reduce(flat)
                        va = acc[T:] / float(dist.get_world_size())
                    else:
                        dist.all_reduce(grad)
                        va = acc[T:].clone()
                        dist.all_reduce(va)
                        va /= float(dist.get_world_size())
                else:
                    va = acc[T:]
                adam_fn(theta, grad, 1.0 / global_T)
                return va
            how = 'one fused call over train + validation tasks (mi_meta_batch_maml_tv: grad_tasks = train tasks), one all-reduce'

        vacc = step_tv()
        nsec = max(2, min(5, args.steps))
        fence()
        t1 = time.perf_counter()
        for _ in range(nsec):
            vacc = step_tv()
        fence()
        dt_tv = (time.perf_counter() - t1) / nsec
        if dist is not None:
            tmax = torch.tensor([dt_tv], device='cuda', dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_tv = tmax.item()
        secondary = {'metric': 'iterations/sec (train + validation halves)', 'value': round(1.0 / dt_tv, 3), 'ms_per_iteration': round(dt_tv * 1e3, 3),
                     'tasks_per_iteration': f'{global_T} train + {global_T} validation', 'steps': nsec, 'how': how,
                     'valid_acc_mean': round(float(vacc.mean()), 5)}
        if not wl.get('anil'):
            del tv

    # Sampled leg: every step DRAWS its meta-batch -- learn2learn TaskDataset semantics (NWays / KShots(2 * shots) /
    # RemapLabels / ConsecutiveLabels, utils/data_pre.py:70-112; the reference calls tasks.sample() per task on the host, maml_vision.py:103) --
    # from a dataset resident in HBM: the host draws image indices, one mi_sample_tasks launch gathers the pixels, then the same step.
    sampled = None
    if not args.no_sampled and not wl.get('anil') and wl['dataset'] == 'min':
        from exploring_meta_amd.utils.task_sampler import ResidentDataset, TaskSampler
        k2 = 2 * wl['shots']
        imgs = torch.cat([d.reshape(-1, *d.shape[2:]) for d, _ in pool[:2]])          # [batches * T * 2SW, C, H, W]: class c = (batch, task, way)
        cls = np.repeat(np.arange(imgs.shape[0] // k2), k2)
        sampler = TaskSampler(ResidentDataset(imgs, cls, device=imgs.device), wl['ways'], wl['shots'], seed=1234 + rank)
        draw = {'host_s': 0.0}

        def compute_sampled(th, _task_ids):
            t_h = time.perf_counter()
            index, lab, rot = sampler.sample_indices(T)
            draw['host_s'] += time.perf_counter() - t_h
            d = sampler.gather(index, rot)
            l = torch.from_numpy(lab).to(d.device)
            loss, acc, grad, _ = run_batch(th, d, l, wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
            return loss, acc, grad

        trainer_smp = MetaTrainer(compute_sampled, adam_fn, global_T)
        keep_theta, keep_adam = theta.clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in adam.items()}
        for _ in range(2):
            trainer_smp.step(theta)
        draw['host_s'] = 0.0
        nsm = max(3, min(10, args.steps))
        fence()
        t_s0 = time.perf_counter()
        for _ in range(nsm):
            sl, sa, _ = trainer_smp.step(theta)
        fence()
        d_sm = (time.perf_counter() - t_s0) / nsm
        if dist is not None:
            tm = torch.tensor([d_sm], device='cuda', dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            d_sm = tm.item()
        sampled = {'metric': 'tasks/sec with the meta-batch drawn every step from a resident dataset', 'value': round(global_T / d_sm, 2),
                   'ms_per_step': round(d_sm * 1e3, 3), 'steps': nsm, 'host_index_draw_ms_per_step': round(draw['host_s'] / nsm * 1e3, 3),
                   'dataset': f'{imgs.shape[0]} images in {imgs.shape[0] // k2} classes ({k2} per class), fp32, resident in HBM',
                   'how': 'TaskSampler.sample_indices on the host (numpy), mi_sample_tasks gathers [T, 2*shots*ways, C, H, W] on the device, '
                          'then the step of the timed region (fused call, all-reduce, Adam)',
                   'query_acc_mean_last_step': round(float(sa), 5)}
        theta.copy_(keep_theta)
        for k, v in keep_adam.items():
            if torch.is_tensor(v):
                adam[k].copy_(v)
            else:
                adam[k] = v
        del imgs, sampler, trainer_smp

    hbm_copy_gbps = measure_stream_copy(eng)
    collective = collective_record(dist, world, theta, eng.param_count + 2 * T,
                                   'one in-place all-reduce per meta-iteration of [meta-gradient | per-task losses | accuracies]')

    # Everything below keeps stepping (other operand form, clock sampling): parameters and optimiser state are put back afterwards, so
    # no theta-dependent figure of this line depends on how long those legs ran.
    snap_theta, snap_adam, snap_n = theta.clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in adam.items()}, out['n']

    def timed(n):
        fence()
        t = time.perf_counter()
        for _ in range(n):
            step()
        fence()
        d = (time.perf_counter() - t) / n
        if dist is not None:
            tm = torch.tensor([d], device='cuda', dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            d = tm.item()
        return d

    # The same step with the hidden convolutions on the exact fp32 matrix pipe (mi_conv_set_split_bf16(0)): the headline's operand
    # form is the split-bf16 one, this is what the arithmetic change is worth, in the same run on the same box.
    fp32_pipe = None
    mask = C.c_uint(0)
    # (pooling nets only: the stride-1 hidden blocks are where the two operand forms differ; a bisecting run's variant mask is left alone)
    if wl['dataset'] == 'min' and not args.no_fp32_pipe and eng.lib.mi_conv_get_split_bf16(C.byref(mask)) and mask.value == 0x3ffff:
        form_was = eng.lib.mi_conv_set_split_bf16(0)
        n32 = max(3, min(5, args.steps))
        timed(2)
        d32 = timed(n32)
        eng.lib.mi_conv_set_split_bf16(form_was)
        fp32_pipe = {'ms_per_step': round(d32 * 1e3, 3), 'tasks_per_s': round(global_T / d32, 2), 'steps': n32,
                     'note': 'the same step with the hidden 3x3 convolutions and weight gradients on the exact fp32 matrix pipe '
                             '(v_mfma_f32_32x32x2_f32, mi_conv_set_split_bf16(0)); measured after the timed region'}
    # The two-plane fp16 operand form (mi_conv_set_split_bf16(2)): 22-bit operands, i.e. narrower than the reference's fp32 however small its
    # measured errors -- a separately labelled, non-default line, never `value`.
    fp16_planes = None
    if wl['dataset'] == 'min' and not args.no_fp32_pipe and eng.lib.mi_conv_get_split_bf16(C.byref(mask)) == 1 and mask.value == 0x3ffff:
        theta.copy_(snap_theta)
        form_was = eng.lib.mi_conv_set_split_bf16(2)
        n16 = max(3, min(5, args.steps))
        timed(2)
        d16 = timed(n16)
        eng.lib.mi_conv_set_split_bf16(form_was)
        fp16_planes = {'ms_per_step': round(d16 * 1e3, 3), 'tasks_per_s': round(global_T / d16, 2), 'steps': n16,
                       'note': 'NOT the headline: the same step with the hidden 3x3 convolutions and weight gradients on the opt-in two-plane fp16 '
                               'operand form (every operand as two fp16 planes scaled by a per-task power of two: 22 bits of each operand, three '
                               'v_mfma_f32_32x32x16_f16 products per multiply-add instead of six bf16 ones; MI_CONV_BF16X3=2); measured after the '
                               'timed region.  Per-kernel errors against fp64 are at or below the fp32 pipe\'s and every parity test runs this '
                               'form too, but its operands are narrower than the reference\'s fp32, so it is reported beside the headline only'}
    # The other reading of north_star's ">= 6x at 8 GPUs" in the same run: with N > 1 ranks under the default weak scaling, a short leg that keeps
    # the GLOBAL meta-batch at the workload's size and gives every rank its contiguous share of it (what --scaling strong times as `value`).
    strong = None
    if world > 1 and args.scaling == 'weak' and Tw >= world and not args.no_secondary:
        lo_s, hi_s = shard_range(Tw, rank, world)
        Ts = hi_s - lo_s
        theta.copy_(snap_theta)

        def compute_strong(th, _task_ids):
            d, l = pool[out['n'] % len(pool)]
            out['n'] += 1
            loss, acc, grad, _ = run_batch(th, d[:Ts], l[:Ts], wl['shots'], wl['steps'], wl['lr'], first_order=wl['first_order'])
            return loss, acc, grad

        trainer_s = MetaTrainer(compute_strong, adam_fn, Tw)
        for _ in range(2):
            trainer_s.step(theta)
        ns = max(3, min(10, args.steps))
        fence()
        t_s = time.perf_counter()
        for _ in range(ns):
            trainer_s.step(theta)
        fence()
        d_s = (time.perf_counter() - t_s) / ns
        tm = torch.tensor([d_s], device='cuda', dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        d_s = tm.item()
        strong = {'scaling': 'strong', 'global_meta_batch': Tw, 'tasks_per_rank': Ts, 'n_gpus': world, 'steps': ns,
                  'ms_per_iteration': round(d_s * 1e3, 3), 'tasks_per_s': round(Tw / d_s, 2),
                  'note': 'the global meta-batch kept at the workload\'s size, each rank its contiguous share, one all-reduce per iteration; '
                          'measured after the timed region (value / ms_per_step of this line are the weak-scaling figures)'}
        if secondary is not None:
            secondary['strong_scaling'] = strong
    clock = clock_record(step, world, ms_per_step=dt / args.steps * 1e3) if rank == 0 and not args.no_clock else None
    theta.copy_(snap_theta)
    for k, v in snap_adam.items():
        if torch.is_tensor(v):
            adam[k].copy_(v)
        else:
            adam[k] = v
    out['n'] = snap_n

    roofline = None
    if dom in prof:
        ms, cnt = prof[dom]
        flops, nbytes = RF.op_costs(spec, dom[0], dom[1], n_img)
        split = int(eng.lib.mi_conv_get_split_bf16(None))      # the operand form of the hidden convolutions (read, not written): 0, 1, 2
        pipe_peak, pipe = RF.mfma_peak(spec, dom[0], dom[1], split)
        bound = RF.bound_of(flops, nbytes, pipe_peak)
        h, w, ci, co, ho, wo, _, _ = RF.layer_geometry(spec)[dom[1]]
        sec = ms / cnt * 1e-3
        if bound == 'mfma':
            achieved, peak, unit = flops / sec / 1e12, round(pipe_peak, 1), 'TFLOP/s'
        else:
            achieved, peak, unit = nbytes / sec / 1e9, HBM_PEAK_GBS, 'GB/s'
        traffic = None
        tpath = os.path.join(REPO, 'profiles', 'pmc_traffic.json')       # HBM bytes/launch from rocprofv3 --pmc passes (profiles/README.md)
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f'{args.workload},{dom[0]},{dom[1]}')
        # every conv-family / streaming kernel of the untimed single-stream step (no concurrency: durations add up to the iteration)
        iso = []
        for (op, layer), (ms1, cnt1) in sorted(first.items(), key=lambda kv: -kv[1][0]):
            c = RF.op_costs(spec, op, layer, n_img) if layer < spec.n_layers else None
            if c is None or len(iso) >= 12:
                continue
            fl, by = c
            pk1, pipe1 = RF.mfma_peak(spec, op, layer, split)
            b1 = RF.bound_of(fl, by, pk1)
            sec1 = ms1 / cnt1 * 1e-3
            a1, p1 = (fl / sec1 / 1e12, pk1) if b1 == 'mfma' else (by / sec1 / 1e9, HBM_PEAK_GBS)
            iso.append(dict(op=op, block=layer + 1, launches=int(cnt1), avg_launch_ms=round(ms1 / cnt1, 4), bound=b1, pipe=pipe1,
                            achieved=round(a1, 1), frac=round(a1 / p1, 3), tflops=round(fl / sec1 / 1e12, 1) if fl else None))
        # (which kernel of the split-bf16 form this launch takes: 16x16x32 MFMAs from 6 tiles per wave on, csrc/conv_b16.h)
        dom_b16 = bool(split == 1 and dom[0] in RF.SPLIT_BF16_OPS and dom[1] >= 1 and
                       RF.conv_kernel_is_b16(n_img // max(T, 1) * h * w, T, co, eng.lib.mi_conv_set_b16(-1)))
        roofline = dict(kernel=f'{RF.kernel_name(spec, dom[0], dom[1], b16=dom_b16)}, block {dom[1] + 1} ({h}x{w}, {ci}->{co} filters)', op=dom[0],
                        bound=bound, achieved=round(achieved, 2), peak=peak, unit=unit, frac=round(achieved / peak, 4), traffic=traffic,
                        launches=int(cnt), avg_launch_ms=round(ms / cnt, 4), flops_per_launch=flops,
                        algorithmic_bytes_per_launch=nbytes, images_per_launch=n_img,
                        hbm_stream_copy_measured_GBps=round(hbm_copy_gbps, 1), pipe=pipe,
                        algorithmic_tflops=round(flops / sec / 1e12, 2), vs_fp32_mfma_peak=round(flops / sec / 1e12 / FP32_MFMA_PEAK_TF, 4),
                        single_stream_step=iso,
                        note='dominant kernel among those that run alone on the chip (dgrad / wgrad of blocks >= 2 share it with each '
                             'other in the timed region); single_stream_step: per-kernel figures of one untimed step with the side '
                             'stream off, HIP events around every launch.  FLOPs are ALGORITHMIC fp32 FLOPs throughout; a kernel on '
                             'the two-plane fp16 operand form executes three fp16 products per fp32 multiply-add, so its peak is the '
                             'dense 16-bit MFMA rate / 3 = 833.3 TFLOP/s (pipe: "fp16 x3"; the three-plane bf16 form: six products, '
                             '416.7, "bf16 x6"), and vs_fp32_mfma_peak says what the same launch is against the fp32 matrix pipe it no '
                             'longer uses')

    if args.breakdown:                          # every rank runs the extra step (it contains the all-reduce); rank 0 writes
        eng.set_overlap(False)                  # one stream: the per-kernel times add up to the iteration
        eng.profile(True)
        step()
        torch.cuda.synchronize()
        full = eng.profile_collect()
        eng.profile(False)
        eng.set_overlap(not args.no_overlap)
        tot = sum(v[0] for v in full.values())
        if rank == 0:
            with open(args.breakdown, 'w') as f:
                f.write(f'# per-kernel HIP-event time of ONE meta-iteration, workload {args.workload} (T={T} tasks), total {tot:.3f} ms\n')
                f.write('op,layer,launches,total_ms,avg_ms,share\n')
                for (op, layer), (ms, cnt) in sorted(full.items(), key=lambda kv: -kv[1][0]):
                    f.write(f'{op},{layer},{cnt},{ms:.4f},{ms / cnt:.4f},{ms / tot:.4f}\n')

    cpu, post = None, {'query_loss_mean_all_tasks_at_init': round(init_loss_mean, 5), 'query_acc_mean_all_tasks_at_init': round(init_acc_mean, 5),
                       'query_loss_mean_last_step': round(loss_after_timed, 5), 'query_acc_mean_last_step': round(acc_after_timed, 5),
                       'task_pool': f'{len(pool)} resident batches of {T} tasks per GPU, rotated every step (each batch is met again only after '
                                    f'{len(pool) - 1} others: the last timed step does not score a batch the parameters were just fitted to)'}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ol, oa = cpu_baseline_vision(wl, spec, theta0)
        n = len(ol)
        ol, oa = np.asarray(ol), np.asarray(oa)
        note = 'engine vs the CPU oracle (fp32) on the same tasks from the same initial parameters'
        if wl['steps'] > 1:
            note += '; with 5 inner steps at lr 0.5 two fp32 runs diverge per task (chaotic inner loop, DESIGN.md section 7)'
        post.update({'compared_tasks': n, 'engine_loss_mean': round(float(eng_loss0[:n].mean()), 5), 'oracle_loss_mean': round(float(ol.mean()), 5),
                     'engine_acc_mean': round(float(eng_acc0[:n].mean()), 5), 'oracle_acc_mean': round(float(oa.mean()), 5),
                     'max_abs_loss_diff_per_task': float(np.max(np.abs(eng_loss0[:n] - ol))),
                     'max_abs_acc_diff_per_task': float(np.max(np.abs(eng_acc0[:n] - oa))), 'note': note})

    if rank != 0:
        return None
    value = global_T * args.steps / dt
    return {
        'metric': 'tasks/sec', 'value': round(value, 2), 'unit': 'tasks/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': wl['name'], 'tasks_per_gpu': T, 'global_meta_batch': global_T, 'ways': wl['ways'],
                   'shots': wl['shots'], 'adapt_steps': wl['steps'], 'inner_lr': wl['lr'],
                   'second_order': not wl['first_order'], 'parallelism': f'task-sharded dp{world}, 1 all-reduce/iter',
                   'task_hardness': HARDNESS[wl['dataset']]},
        'post_adapt': post, 'secondary': secondary, 'sampled': sampled, 'hbm_stream_copy_GBps': round(hbm_copy_gbps, 1), 'roofline': roofline,
        'cpu_baseline': cpu, 'collective': collective, 'arithmetic': arithmetic_note(eng, T), 'fp32_pipe': fp32_pipe, 'fp16_planes': fp16_planes, 'strong_scaling': strong, 'clock': clock,
    }


def clock_record(step, world, seconds=3.0, ms_per_step=None):
    """Shader clock and socket power while the step loop keeps running (untimed, after the timed region; rocm-smi is a child
    process that only reads).  `roofline.peak` assumes the nominal 2400 MHz; at its socket power cap the MI355X runs the conv
    workloads near 2000 MHz (profiles/r3/clock), so read `roofline.frac` with this clock next to it.  None when world > 1 (the
    extra steps would have to be agreed between ranks) or when rocm-smi is not on PATH."""
    import re
    import shutil
    import subprocess
    if world != 1 or not shutil.which('rocm-smi'):
        return None
    # under rocprofv3 every child carries the profiler's preloaded library, which initialises the GPU before the child's own exec
    # (rocm-smi is a '#!/usr/bin/env python3' script): no child processes from a profiled run
    if 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return None
    samples, t_end = [], time.perf_counter() + seconds
    torch.cuda.synchronize()
    n_steps, t_loop = [0], time.perf_counter()
    step_inner = step

    def step():                                              # (counted: the loop below is also the line's `sustained` figure)
        step_inner()
        n_steps[0] += 1
    try:
        while time.perf_counter() < t_end:
            with subprocess.Popen(['rocm-smi', '-c', '-P'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) as p:
                t_child = time.perf_counter()
                while p.poll() is None:
                    step()
                    if time.perf_counter() - t_child > 5.0:      # a reading takes ~0.1 s: a stuck tool must not hold the bench
                        p.kill()
                        p.wait()
                        t_end = 0.0
                        break
                txt = p.stdout.read() if p.returncode == 0 else ''
            m, w = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', txt), re.search(r'Power \(W\): ([0-9.]+)', txt)
            if m:
                samples.append((int(m.group(1)), float(w.group(1)) if w else None))
            elif not samples:
                break                                        # no reading from this tool on this box: do not keep trying
    except OSError:
        pass
    torch.cuda.synchronize()
    loop_s = time.perf_counter() - t_loop
    samples = samples[1:] or samples                     # the first sample may still see the clock ramp
    if not samples:
        return None
    sclk = sorted(c for c, _ in samples)[len(samples) // 2]
    watts = [x for _, x in samples if x is not None]
    power = sorted(watts)[len(watts) // 2] if watts else None
    return {'sclk_mhz': sclk, 'nominal_mhz': 2400, 'socket_power_w': power,
            'joules_per_step': (round(power * ms_per_step * 1e-3, 3) if power and ms_per_step else None),    # at the cap, time follows energy
            'samples': len(samples), 'how': 'rocm-smi -c -P while the step loop runs, after the timed region',
            # the same step loop over >= 3 s (an observer that samples the GPU every few seconds sees this leg; the headline's timed region is the
            # K steps the command line asks for)
            'sustained': {'steps': n_steps[0], 'seconds': round(loop_s, 3), 'ms_per_step': round(loop_s / max(1, n_steps[0]) * 1e3, 3)}}


def arithmetic_note(eng, tasks_per_call=32):
    """What "f32" means for this line: inputs, outputs, accumulation and every stored tensor are fp32; with the split operand form
    (the default) the hidden convolutions and weight gradients form each fp32 product from exact three-way bf16 splits of both operands."""
    split = int(eng.lib.mi_conv_get_split_bf16(None))
    if not split:
        return {'split_bf16_operands': False, 'operand_form': 'fp32',
                'note': 'fp32 throughout (fp32-input MFMA = an fmaf chain; fp64 for statistics and reductions)'}
    if split == 2:
        return {'split_bf16_operands': False, 'operand_form': 'fp16 x2 planes, 3 products',
                'note': 'fp32 tensors and fp32 accumulation throughout (fp64 for statistics and reductions).  Hidden 3x3 convolutions and weight '
                        'gradients: each fp32 operand, scaled by a power of two chosen per task and tensor from its largest magnitude, as two '
                        'fp16 planes (x s = h + l to 2^-22, 2^-39 of the maximum absolutely), three fp16 MFMA products per multiply-add, the '
                        'scales divided out of the fp32 sums exactly: per-kernel errors against the fp64 oracle are at or below those of the '
                        'fp32 matrix pipe (tests run all three forms against the same bars); MI_CONV_BF16X3=1 selects three exact bf16 '
                        'planes (six products), MI_CONV_BF16X3=0 the fp32 pipe'}
    return {'split_bf16_operands': True, 'operand_form': 'bf16 x3 planes, 6 products',
            'note': 'fp32 tensors and fp32 accumulation throughout (fp64 for statistics and reductions).  Hidden 3x3 convolutions and weight '
                    'gradients: each fp32 operand as the EXACT sum of three bf16 pieces, six bf16 MFMA products per multiply-add, dropped '
                    'cross terms <= 2^-24 of a product (one fp32 rounding): per-kernel errors against the fp64 oracle are the same or smaller '
                    'than with the fp32 matrix pipe (tests run both forms against the same bars); MI_CONV_BF16X3=0 selects the fp32 pipe.  '
                    'Block 1 (three input channels) in the same form since round 6: conv1 of its forward / tangent-forward kernels with eight '
                    'products (the two 2^-24 cross terms kept: raw-pixel inputs), the sparse part of its weight gradient with six'}


def collective_record(dist, world, theta, numel, what):
    """Evidence that N ranks really reduced: the world size RCCL reports, the timed all-reduce of a buffer of
    the step's own size (20 calls after 3 warm-ups, events on the current stream), and a bit-level checksum of the parameters after
    the last timed step gathered from every rank (identical = every rank applied the same reduced gradient)."""
    if dist is None:
        return None
    buf = torch.zeros(numel, dtype=torch.float32, device='cuda')
    for _ in range(3):
        dist.all_reduce(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    chk = theta.detach().contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    sums = [int(c.item()) for c in allc]
    if len(set(sums)) != 1:
        raise SystemExit(f'parameters differ between ranks after the timed steps: checksums {sums}')
    ver = None
    try:
        ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        pass
    backend = str(dist.get_backend())
    return dict(backend=backend + (' (= RCCL on ROCm)' if backend == 'nccl' else ' (not RCCL: MI_DIST_BACKEND rehearsal)'),
                rccl_version=ver if backend == 'nccl' else None, world_size=dist.get_world_size(), what=what,
                allreduce_numel=int(numel), allreduce_bytes=int(numel) * 4, allreduce_us=round(us, 2), timed_calls=20,
                theta_checksum=sums[0], theta_checksum_identical_on_all_ranks=True)


def measure_stream_copy(eng_or_lib):
    """Measured HBM roofline: the build's own streaming-copy kernel, same process, same run (read + write bytes / time)."""
    import ctypes as C
    lib = getattr(eng_or_lib, 'lib', eng_or_lib)
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device='cuda').zero_()
    dst = torch.empty_like(src)
    cp = lambda: lib.mi_stream_copy(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(src.data_ptr()),
                                    C.c_void_p(dst.data_ptr()), nbytes)
    cp()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cp()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


# ---------------------------------------------------------------------------------------------------------------------
TRPO_PARAMS = dict(inner_lr=0.1, max_path_length=100, adapt_steps=1, adapt_batch_size=20, meta_batch_size=20, outer_lr=0.3,
                   backtrack_factor=0.5, ls_max_steps=15, max_kl=0.01, tau=1.0, gamma=0.99)


def run_trpo(args, wl, rank, world, local, dist):
    """BASELINE config 5: one step = meta_optimize_trpo (rl/maml_trpo.py:130-134; core_functions/rl.py:409-473) over the resident
    replays of this rank's tasks: host GAE / baseline fits, one surrogate + gradient call, 11 Fisher-vector products inside CG,
    up to 15 line-search evaluations -- every device call covers all local tasks.  Rollouts (env stepping) are data generation and
    happen once, before the timed region, through the package's own fast_adapt_trpo."""
    from copy import deepcopy
    from exploring_meta_amd import core_functions as cf
    p = dict(TRPO_PARAMS)
    Tw = wl['tasks']
    if args.scaling == 'strong':
        lo, hi = shard_range(Tw, rank, world)
        global_T = Tw
    else:
        lo, hi, global_T = rank * Tw, (rank + 1) * Tw, Tw * world
    p['meta_batch_size'] = global_T
    dev = torch.device('cuda', local)
    cf.set_device(dev)
    torch.manual_seed(42)
    policy = cf.DiagNormalPolicy(2, 2).to(dev)
    baseline = cf.LinearValue(2, 2)
    goals = np.random.RandomState(42).uniform(-0.5, 0.5, size=(global_T, 2))
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    replays, olds = [], []
    for goal in goals[lo:hi]:
        learner = deepcopy(policy)
        task = cf.Particles2DRunner(goal, p['max_path_length'], gen, dev)
        learner, _, rep, _, _ = cf.fast_adapt_trpo(task, learner, baseline, p, first_order=True)
        replays.append(rep)
        olds.append(learner)
    theta0 = policy.flat().clone()
    out = {}

    def step():
        policy.load_flat(theta0)                # every step optimises from the same parameters: identical work per step
        out['r'] = cf.meta_optimize_trpo(p, policy, baseline, replays, olds)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    # Dominant device call: the Fisher-vector product (11 per step).  Timed with events on the stream it is launched on.
    ctx = out['r']['context']
    v = torch.randn_like(theta0)
    ctx.evaluate(theta0, want_grad=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nf = 50
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nf):
        ctx.fvp(theta0, v)
    e1.record()
    torch.cuda.synchronize()
    fvp_ms = e0.elapsed_time(e1) / nf
    B = int(ctx.qry['states'].shape[1])
    fwd = 2.0 * B * (2 * 100 + 100 * 100 + 100 * 2)                   # one dense forward over a task's padded batch
    flops = 16.0 * fwd * (hi - lo)                                       # (I - aH) F (I - aH) v: two 6-pass Hessian products + JVP/VJP through the query forward
    achieved = flops / (fvp_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):            # HBM bytes of ONE Hessian-vector sweep launch (the dominant kernel of a product: two of its three sweeps)
        traffic = json.load(open(tpath)).get('cfg5,fisher_vector_product,0')
    roofline = dict(kernel='mi_trpo_fvp = 3 fused sweeps (policy_sweep_kernel: H_t v over the support pass, F_t u over the query pass, H_t w over the '
                           'support pass) + 3 folds, the last of which also takes the mean over tasks: 6 launches',
                    op='fisher_vector_product', bound='mfma', achieved=round(achieved, 3), peak=FP32_MFMA_PEAK_TF, unit='TFLOP/s',
                    frac=round(achieved / FP32_MFMA_PEAK_TF, 5), traffic=traffic, launches=nf, avg_launch_ms=round(fvp_ms, 4),
                    flops_per_launch=flops, traffic_note='counter bytes of one policy_sweep_kernel<HVP> launch (a product = 2 such sweeps + 1 Fisher sweep)',
                    note='2x100 MLP on 2000-row batches; algorithmic FLOPs = 16 dense forward passes of a task batch per product '
                    '(two 6-pass Hessian-vector sweeps + tangent forward / backward through the query pass); a sweep takes 32-row slabs through the whole '
                    'chain inside one workgroup with both 100x100 weight matrices resident in LDS (csrc/policy_sweep.h); round 2 ran one product as ~34 '
                    'per-layer launches in 0.86 ms')

    collective = collective_record(dist, world, policy.flat(), theta0.numel() + 2,
                                   'task-count-weighted means of (loss, KL, gradient) and of every Fisher-vector product: one all-reduce each')
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_trpo(p, policy, theta0, replays, olds, out['r'])
    if rank != 0:
        return None
    value = global_T * args.steps / dt
    r = out['r']
    return {
        'metric': 'tasks/sec', 'value': round(value, 2), 'unit': 'tasks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': wl['name'], 'tasks_per_gpu': hi - lo, 'global_meta_batch': global_T, 'rows_per_replay': B,
                   'parallelism': f'task-sharded dp{world}, 1 + 11 + <=15 small all-reduces per step',
                   **{k: p[k] for k in ('inner_lr', 'max_kl', 'adapt_steps')}},
        'post_adapt': {'surrogate_loss_before': float(r['old_loss']), 'surrogate_loss_after': None if r['new_loss'] is None else float(r['new_loss']),
                       'kl_after': None if r['kl'] is None else float(r['kl']), 'line_search_step': r['accepted'],
                       **(cpu.pop('_post') if cpu else {})},
        'secondary': {'metric': 'meta_optimize_trpo iterations/sec', 'value': round(args.steps / dt, 3)},
        'roofline': roofline, 'cpu_baseline': cpu, 'collective': collective,
        'clock': clock_record(step, world, ms_per_step=dt / args.steps * 1e3) if not args.no_clock else None,
    }


def cpu_baseline_trpo(p, policy, theta0, replays, olds, gpu_out):
    """oracle/rl_ref.py::meta_optimize_trpo (fp64 autograd restatement of rl.py:409-473) on the SAME replays and old policies."""
    from collections import OrderedDict
    from oracle import rl_ref as RL

    def named(flat):
        outp, off = OrderedDict(), 0
        for k, prm in policy.named_parameters():
            n = prm.numel()
            outp[k] = flat[off:off + n].reshape(prm.shape).detach().cpu().double().clone()
            off += n
        return outp

    cpu_replays = [[{k: v.detach().cpu().double() for k, v in rep.items()} for rep in task] for task in replays]
    cpu_olds = [named(o.flat()) for o in olds]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    t0 = time.perf_counter()
    p64 = OrderedDict((k, v.requires_grad_(True)) for k, v in named(theta0).items())
    ref = RL.meta_optimize_trpo(p, p64, RL.LinearValue(2, 2), cpu_replays, cpu_olds)
    dt = time.perf_counter() - t0
    gstep = gpu_out['step'].double().cpu()
    rel = float((gstep - ref['step']).norm() / ref['step'].norm())
    return dict(value=len(replays) / dt, unit='tasks/s', cores=torch.get_num_threads(), kind='port', cpu=cpu_model(),
                host_logical_cores=os.cpu_count(),
                sample=f'one meta_optimize_trpo over the same {len(replays)} tasks (same replays, same old policies), '
                       f'oracle/rl_ref.py fp64 autograd, {torch.get_num_threads()} intra-op threads',
                _post={'oracle_line_search_step': ref['accepted'], 'step_direction_rel_err_vs_oracle': rel})


# ---------------------------------------------------------------------------------------------------------------------
def other_workloads(current, steps=5, warmup=2):
    """The other BASELINE configurations, 5 timed steps each in a child process of this run (same box, same build, after the headline's timed
    region): {ms_per_step, tasks_per_s, the dominant kernel's roofline fraction}.  The headline's `value` is untouched by this."""
    import subprocess
    out = {}
    for name in sorted(WORKLOADS):
        if name == current:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', name, '--steps', str(steps), '--warmup', str(warmup), '--pool', '2',
               '--no-cpu-baseline', '--no-fp32-pipe', '--no-secondary', '--no-clock', '--no-dist', '--no-other', '--no-sampled']
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150, env=env)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d.get('roofline') or {}
            out[name] = {'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'tasks_per_s': d['value'], 'steps': d['steps'],
                         'roofline': {'kernel': rf.get('kernel'), 'bound': rf.get('bound'), 'frac': rf.get('frac'), 'avg_launch_ms': rf.get('avg_launch_ms')},
                         'wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:                               # a failed side run must not take the headline with it
            out[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--tasks', type=int, default=0, help='override the tasks per GPU of the workload (sweeps; not a BASELINE configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-clock', action='store_true', help='skip the shader-clock / socket-power record (3 s of extra untimed steps)')
    ap.add_argument('--graph', type=int, default=-1, help='replay the fused call as a captured hipGraph (mi_engine_set_graph): 1 on, 0 off, '
                    '-1 = the workload default (on for the launch-bound few-image configurations cfg1 and cfg4)')
    ap.add_argument('--breakdown', default='', help='write a per-kernel event-time breakdown (one extra untimed step) to this file')
    ap.add_argument('--pool', type=int, default=8, help='distinct resident task batches the step loop rotates through (vision workloads)')
    ap.add_argument('--no-overlap', action='store_true', help='weight gradients on the main stream too (mi_engine_set_overlap(0)): per-launch '
                    'durations in a kernel trace are then those of kernels running alone')
    ap.add_argument('--no-secondary', action='store_true', help='skip the train + validation leg (counter passes: its fused call launches the same '
                    'kernels with twice the tasks, which a per-kernel counter table cannot tell from the timed workload)')
    ap.add_argument('--no-fp32-pipe', action='store_true', help='skip the fp32-pipe leg (counter passes: only the shipped operand form is launched)')
    ap.add_argument('--no-dist', action='store_true', help='N = 1 without the single-rank process group (profiler runs)')
    ap.add_argument('--no-sampled', action='store_true', help='skip the leg that draws every step\'s meta-batch with mi_sample_tasks from a resident dataset')
    ap.add_argument('--no-other', action='store_true', help='skip the short runs of the other BASELINE configurations (N = 1 only; child processes)')
    ap.add_argument('--launch-check', action='store_true', help='ranks only join the process group, all-reduce one number and rank 0 prints '
                    '{"launch_check": true, "world_size": N}: exercises the self-launch path without a GPU (MI_DIST_BACKEND=gloo)')
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.tasks:
        wl['tasks'] = args.tasks
        wl['name'] += f' [tasks per GPU overridden: {args.tasks}]'

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:             # (a bare `python bench.py --gpus N` never gets here: self_launch started the ranks)
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree')
    gloo = os.environ.get('MI_DIST_BACKEND', 'nccl') != 'nccl'
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group('gloo' if gloo or not torch.cuda.is_available() else 'nccl')
        one = torch.ones(1) if dist.get_backend() == 'gloo' else torch.ones(1, device=torch.device('cuda', local))
        dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({'launch_check': True, 'world_size': dist.get_world_size(), 'allreduce_of_ones': float(one.item()),
                              'backend': str(dist.get_backend())}), file=_RESULT_OUT or sys.stdout, flush=True)
        dist.destroy_process_group()
        return
    if gloo and torch.cuda.device_count() < world:
        local = local % max(1, torch.cuda.device_count())       # rehearsal of N ranks on fewer cards (collective over gloo)
    torch.cuda.set_device(local)
    dist, dist_note = None, None
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run always go through RCCL (also exercised at N=1)
        import torch.distributed as dist
        init_process_group(local)               # RCCL; MI_DIST_BACKEND=gloo only to rehearse ranks that share a card
    elif not args.no_dist and not profiled:
        # plain `python bench.py` (the driver's N = 1 command): a single-rank RCCL group in this process, so that the step takes the
        # same all-reduce path as N > 1 and the line carries a `collective` record
        import torch.distributed as dist
        os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()), 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': str(local)})
        try:
            init_process_group(local)
        except Exception as e:                  # the headline does not depend on the collective library at N = 1: say so and go on
            dist, dist_note = None, f'single-rank process group failed: {type(e).__name__}: {e}'
    runner = run_trpo if wl.get('kind') == 'trpo' else run_vision
    line = runner(args, wl, rank, world, local, dist)
    if rank == 0 and world == 1 and line is not None and not args.no_other and not profiled and not args.tasks:
        line['other_workloads'] = other_workloads(args.workload)
    if rank == 0:
        if dist_note and line is not None:
            line['collective'] = {'error': dist_note}
        print(json.dumps(line), file=_RESULT_OUT or sys.stdout, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
