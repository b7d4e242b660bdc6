"""Child process of tests/test_gpu_two_ranks.py: one rank of a 2-rank job whose ranks share cuda:0 (the GPU box has one card, and
RCCL refuses two ranks on one GPU, so the collective goes over gloo: MI_DIST_BACKEND=gloo).  Everything else is the product's N > 1
path: shard_range -> the rank's own MetaEngine -> one all-reduce -> the same Adam step on every rank.

    python two_rank_worker.py <mode> <out.pt>      with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

TRAINER = dict(dataset='omni', ways=5, shots=1, steps=2, lr=0.4, tasks=6, seed=13)
DRIVER = dict(ways=5, shots=1, adapt_steps=1, meta_batch_size=4, num_iterations=2, inner_lr=0.4, save_every=1)


def trainer_step(cfg=TRAINER):
    """One task-sharded meta-iteration on the real engine, bench.py's arrangement (run_vision): returns what the step produced."""
    import torch.distributed as dist
    from exploring_meta_amd.engine import MetaEngine, ModelSpec
    from exploring_meta_amd.sharding import MetaTrainer
    from exploring_meta_amd.utils import synthetic
    spec = ModelSpec.omniglot(cfg['ways'])
    eng = MetaEngine(spec)
    shapes = dict(spec.param_shapes())
    w = synthetic.ref_init_weights(shapes, cfg['seed'])
    theta = torch.from_numpy(np.concatenate([w[k].ravel() for k in shapes])).float().cuda()
    adam, seen = {}, {'grads': []}

    def compute(th, task_ids):
        d, l = synthetic.make_meta_batch(cfg['dataset'], task_ids, cfg['ways'], cfg['shots'])
        loss, acc, grad, _ = eng.meta_batch(th, torch.from_numpy(d).cuda(), torch.from_numpy(l).cuda(), cfg['shots'], cfg['steps'], cfg['lr'],
                                            first_order=False)
        seen['local_tasks'] = list(task_ids)
        return loss, acc, grad

    def adam_fn(th, grad, scale):
        seen['grads'].append(grad.detach().clone().cpu())
        eng.adam_step(th, grad, adam, 0.003, grad_scale=scale)

    trainer = MetaTrainer(compute, adam_fn, cfg['tasks'])
    loss, acc, _ = trainer.step(theta)
    loss2, acc2, _ = trainer.step(theta, first_task_id=cfg['tasks'])          # a second iteration on the updated parameters
    torch.cuda.synchronize()
    return dict(grad=seen['grads'][0], grad2=seen['grads'][1], loss=float(loss), acc=float(acc), loss2=float(loss2), acc2=float(acc2), theta=theta.cpu(),
                local_tasks=seen['local_tasks'], world=dist.get_world_size() if dist.is_initialized() else 1)


def driver_run(save_dir):
    from exploring_meta_amd.vision import maml_vision
    logs = []
    p = dict(maml_vision.params, **DRIVER, save_dir=save_dir)
    model, metrics = maml_vision.run('omni', p, first_order=False, log=logs.append)
    return dict(sd={k: v.cpu() for k, v in model.state_dict().items()}, metrics=metrics, logs=logs)


def trpo_run():
    from exploring_meta_amd.rl import maml_trpo
    logs = []
    p = dict(maml_trpo.params, meta_batch_size=4, num_iterations=2, adapt_batch_size=4, adapt_steps=1, max_path_length=20, seed=7)
    policy = maml_trpo.run(p, log=logs.append)
    return dict(sd={k: v.cpu() for k, v in policy.state_dict().items()}, logs=logs)


if __name__ == '__main__':
    mode, out = sys.argv[1], sys.argv[2]
    torch.set_num_threads(4)
    if mode == 'trainer':
        from exploring_meta_amd.sharding import init_process_group
        local = int(os.environ.get('LOCAL_RANK', '0'))       # 0 for both ranks where they share the card (gloo), the rank's own GPU under RCCL
        torch.cuda.set_device(local)
        init_process_group(local)
        res = trainer_step()
        torch.distributed.destroy_process_group()
    elif mode == 'trpo':
        res = trpo_run()
    else:
        res = driver_run(os.path.join(os.path.dirname(out), f'ckpt_rank{os.environ["RANK"]}'))
    torch.save(res, out)
