"""Task-sharded data parallelism for the meta-batch: tasks are independent given the meta-parameters
(reference vision/maml_vision.py:102-124), so rank r owns a contiguous block of the global task list, runs them through its
own engine, and ONE all-reduce of the flat fp32 meta-gradient (with the loss/accuracy sums appended, so logging needs no
second collective) precedes the identical Adam step on every rank.  Backend "nccl" is RCCL over xGMI on the GPU box;
"gloo" on CPU (tests).  No collective touches the per-task data path.
"""
import torch
import torch.distributed as dist


def init_process_group(local_rank):
    """Join the job `torchrun` described in the environment.  Backend "nccl" (= RCCL) with the rank's device bound up front;
    ``MI_DIST_BACKEND=gloo`` is for rehearsing N > 1 where ranks must share a card (RCCL refuses two ranks on one GPU): the
    engine path is the same, only the all-reduce is staged through the host."""
    import os
    backend = os.environ.get('MI_DIST_BACKEND', 'nccl')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend)


def shard_range(num_tasks, rank, world):
    """Contiguous block of the global task list owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(num_tasks, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_meta_batch(meta_grad, loss_sum, acc_sum, group=None, extra=()):
    """Sum (meta_grad, loss_sum, acc_sum, *extra) over ranks with a single all-reduce: the scalars (train loss / accuracy sums and
    whatever else the caller logs, e.g. the validation sums) and any small tensors (e.g. the BatchNorm running-statistics
    contribution of this rank's forward passes) ride behind the gradient.  Returns the reduced (meta_grad, loss_sum, acc_sum) --
    with ``extra`` a 4-tuple whose last element is the list of reduced extras, each in the shape it was given.
    Single-process (no initialised process group): identity."""
    extra = list(extra)
    if not (dist.is_available() and dist.is_initialized()):
        return (meta_grad, loss_sum, acc_sum, extra) if extra else (meta_grad, loss_sum, acc_sum)
    tail = [torch.as_tensor(x, device=meta_grad.device).to(meta_grad.dtype) for x in [loss_sum, acc_sum] + extra]
    flat = torch.cat([meta_grad.reshape(-1)] + [t.reshape(-1) for t in tail])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    parts, off = [], meta_grad.numel()
    for t in tail:
        parts.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    out = (flat[:meta_grad.numel()].view_as(meta_grad), parts[0].reshape(()), parts[1].reshape(()))
    return out + (parts[2:],) if extra else out


def packed_outputs(grad, loss, acc):
    """If (grad, loss, acc) are adjacent views [grad | loss | acc] of one allocation (how MetaEngine.meta_batch returns them), the
    flat view over all three -- the all-reduce then needs no gather launch -- else None."""
    try:
        same = (grad.untyped_storage().data_ptr() == loss.untyped_storage().data_ptr() == acc.untyped_storage().data_ptr())
    except (AttributeError, RuntimeError):
        return None
    if not same or grad.dtype != loss.dtype or grad.dtype != acc.dtype or not (grad.is_contiguous() and loss.is_contiguous() and acc.is_contiguous()):
        return None
    if loss.dim() != 1 or acc.shape != loss.shape or loss.storage_offset() != grad.storage_offset() + grad.numel() or \
            acc.storage_offset() != loss.storage_offset() + loss.numel():
        return None
    return torch.as_strided(grad, (grad.numel() + 2 * loss.numel(),), (1,), grad.storage_offset())


class MetaTrainer:
    """One meta-iteration = local shard through ``compute`` -> one all-reduce -> Adam (maml_vision.py:93-141, train half).

    ``compute(theta, task_ids) -> (loss[T_local], acc[T_local], meta_grad_sum[P])`` is the engine call on this rank's shard
    (``MetaEngine.meta_batch`` on the GPU; tests inject a CPU callable).  ``adam(theta, grad, grad_scale)`` applies the step
    in place.  theta is replicated; since every rank applies the same step to the same reduced gradient it stays identical
    without a broadcast.
    """

    def __init__(self, compute, adam, meta_batch_size, group=None):
        self.compute, self.adam, self.meta_batch_size, self.group = compute, adam, meta_batch_size, group
        init = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if init else 0
        self.world = dist.get_world_size(group) if init else 1

    def local_tasks(self, first_task_id=0):
        a, b = shard_range(self.meta_batch_size, self.rank, self.world)
        return list(range(first_task_id + a, first_task_id + b))

    def step(self, theta, first_task_id=0):
        tasks = self.local_tasks(first_task_id)
        if tasks and not (dist.is_available() and dist.is_initialized()):
            # no process group (plain `python bench.py`): the two means in one stacked reduction (each tiny reduction / division launch is a measurable share
            # of the few-image configurations' 0.6..1.8 ms iterations)
            loss, acc, grad = self.compute(theta, tasks)
            flat = packed_outputs(grad, loss, acc)
            both = flat[grad.numel():].view(2, -1) if flat is not None else torch.stack((loss, acc))
            means = both.sum(dim=1) / self.meta_batch_size
            self.adam(theta, grad, 1.0 / self.meta_batch_size)        # maml_vision.py:139-141
            return means[0], means[1], grad
        if tasks:
            loss, acc, grad = self.compute(theta, tasks)
            flat = packed_outputs(grad, loss, acc) if self.meta_batch_size % self.world == 0 else None
            if flat is not None:
                # equal shards and the engine's packed outputs: ONE all-reduce of [grad | loss | acc] in place, no gather / sum
                # launches in front of it (the per-task losses of different ranks add up elementwise: only their total is used)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                means = flat[grad.numel():].view(2, -1).sum(dim=1) / self.meta_batch_size
                self.adam(theta, grad, 1.0 / self.meta_batch_size)    # maml_vision.py:139-141
                return means[0], means[1], grad
            loss_sum, acc_sum = loss.sum(), acc.sum()
        else:      # meta-batch smaller than the world: this rank owns no task and contributes zeros to the all-reduce
            grad = torch.zeros_like(theta)
            loss_sum = acc_sum = torch.zeros((), dtype=theta.dtype, device=theta.device)
        grad, loss_sum, acc_sum = reduce_meta_batch(grad, loss_sum, acc_sum, self.group)
        self.adam(theta, grad, 1.0 / self.meta_batch_size)            # maml_vision.py:139-141
        return loss_sum / self.meta_batch_size, acc_sum / self.meta_batch_size, grad
