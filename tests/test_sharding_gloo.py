"""N>1 path on CPU: world_size-2 gloo run of the task-sharded meta-iteration (sharding.MetaTrainer) must give the same
reduced meta-gradient, metrics and Adam-updated parameters as the single-process run over the whole meta-batch.  The local
engine call is replaced by the oracle here (no GPU in this container); the collective/sharding/optimizer plumbing under test
is the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _compute_factory():
    from collections import OrderedDict
    from exploring_meta_amd.utils import synthetic
    from oracle import vision_ref as R
    spec = R.omniglot_spec(5)
    shapes = R.param_shapes(spec)

    def compute(theta, task_ids):
        th, off = OrderedDict(), 0
        for k, shp in shapes.items():
            n = int(np.prod(shp))
            th[k] = theta[off:off + n].view(shp).double()
            off += n
        datas, labels = [], []
        for t in task_ids:
            d, l = synthetic.make_task('omni', t, 5, 1)
            datas.append(torch.from_numpy(d).double())
            labels.append(torch.from_numpy(l))
        loss, acc, grad, _ = R.maml_meta_batch(th, spec, datas, labels, 1, 1, 5, 0.5, True)
        return loss, acc.double(), R.flatten_params(grad)

    w = synthetic.hash_weights(shapes, 11)
    theta0 = torch.cat([torch.from_numpy(v).reshape(-1) for v in w.values()])
    return compute, theta0, R


def _run(rank, world, port, out_path, meta_batch, packed=False):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch.distributed as dist
    from exploring_meta_amd.sharding import MetaTrainer
    torch.set_num_threads(2 if world <= 2 else 1)
    if world > 1:
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    compute, theta, R = _compute_factory()
    if packed:      # the engine's output layout: [grad | loss | acc] views of ONE allocation (MetaEngine._outputs)
        inner = compute

        def compute(th, task_ids):
            loss, acc, grad = inner(th, task_ids)
            buf = torch.cat([grad.reshape(-1), loss.reshape(-1).to(grad.dtype), acc.reshape(-1).to(grad.dtype)])
            P, T = grad.numel(), loss.numel()
            return buf[P:P + T], buf[P + T:], buf[:P]
    m, v, step = torch.zeros_like(theta), torch.zeros_like(theta), [0]

    def adam(th, grad, scale):
        step[0] = R.adam_step(th, grad * scale, m, v, step[0])

    tr = MetaTrainer(compute, adam, meta_batch)
    assert len(tr.local_tasks()) in (meta_batch // world, meta_batch // world + 1)
    outs = []
    for it in range(2):
        loss, acc, grad = tr.step(theta, first_task_id=it * meta_batch)
        outs.append((float(loss), float(acc), grad.clone()))
    if rank == 0:
        torch.save(dict(theta=theta, outs=outs), out_path)
    if world > 1:
        # every rank must hold bit-identical parameters after the steps
        gathered = [torch.zeros_like(theta) for _ in range(world)]
        dist.all_gather(gathered, theta)
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        dist.destroy_process_group()


@pytest.mark.parametrize('meta_batch', [4, 33, 1])     # even split; ragged 17 + 16; rank 1 owns no task
def test_two_ranks_match_single_process(tmp_path, meta_batch):
    single = str(tmp_path / 'single.pt')
    _run(0, 1, 0, single, meta_batch)
    port = _free_port()
    multi = str(tmp_path / 'multi.pt')
    mp.spawn(_run, args=(2, port, multi, meta_batch), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(multi)
    for (la, aa, ga), (lb, ab, gb) in zip(a['outs'], b['outs']):
        assert la == pytest.approx(lb, rel=1e-12) and aa == pytest.approx(ab, rel=1e-12)
        assert float((ga - gb).norm() / ga.norm()) < 1e-12        # fp64 sums in a different order (per-rank partial sums)
    # Adam normalises by sqrt(v): where the oracle's autograd gradient is pure rounding noise (conv biases under batch-stat
    # BN, ~1e-17) the sign of the step is arbitrary, so parameters are compared where the gradient is above noise.
    g = a['outs'][-1][2].abs()
    live = g > 1e-9 * g.max()
    assert live.float().mean() > 0.9
    assert torch.allclose(a['theta'][live], b['theta'][live], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('meta_batch,packed', [(32, True), (20, False)])
def test_eight_ranks_match_single_process(tmp_path, meta_batch, packed):
    """World size 8 (the node the scaling bench runs on; gloo here): a 32-task meta-batch (4 tasks per rank: the in-place all-reduce of the
    packed [grad | loss | acc] buffer) and a 20-task one (config 5's split over 8 ranks: 3/3/3/3/2/2/2/2, the gather path).  theta is
    bit-identical on all eight ranks after two iterations (asserted inside every rank) and equal to the single-process run up to the order
    of the fp64 sums."""
    from exploring_meta_amd.sharding import shard_range
    sizes = [b - a for a, b in (shard_range(meta_batch, r, 8) for r in range(8))]
    assert sizes == ([4] * 8 if meta_batch == 32 else [3, 3, 3, 3, 2, 2, 2, 2])
    single, multi = str(tmp_path / 'single.pt'), str(tmp_path / 'multi.pt')
    _run(0, 1, 0, single, meta_batch, packed)
    mp.spawn(_run, args=(8, _free_port(), multi, meta_batch, packed), nprocs=8, join=True)
    a, b = torch.load(single), torch.load(multi)
    for (la, aa, ga), (lb, ab, gb) in zip(a['outs'], b['outs']):
        assert la == pytest.approx(lb, rel=1e-12) and aa == pytest.approx(ab, rel=1e-12)
        assert float((ga - gb).norm() / ga.norm()) < 1e-12
    g = a['outs'][-1][2].abs()
    live = g > 1e-9 * g.max()
    assert torch.allclose(a['theta'][live], b['theta'][live], rtol=1e-9, atol=1e-12)


def test_packed_outputs_take_the_in_place_all_reduce(tmp_path):
    """Equal shards + the engine's packed [grad | loss | acc] outputs: MetaTrainer all-reduces that one buffer in place (no gather /
    sum launches in front of the collective).  Same results as the single-process run and as the gather path."""
    from exploring_meta_amd.sharding import packed_outputs
    buf = torch.arange(10.0)
    flat = packed_outputs(buf[:6], buf[6:8], buf[8:])
    assert flat is not None and flat.data_ptr() == buf.data_ptr() and flat.numel() == 10
    assert packed_outputs(buf[:6].clone(), buf[6:8], buf[8:]) is None and packed_outputs(buf[:5], buf[6:8], buf[8:]) is None
    single, multi = str(tmp_path / 'single.pt'), str(tmp_path / 'multi.pt')
    _run(0, 1, 0, single, 4, True)
    mp.spawn(_run, args=(2, _free_port(), multi, 4, True), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(multi)
    for (la, aa, ga), (lb, ab, gb) in zip(a['outs'], b['outs']):
        assert la == pytest.approx(lb, rel=1e-12) and aa == pytest.approx(ab, rel=1e-12)
        assert float((ga - gb).norm() / ga.norm()) < 1e-12


def test_shard_range_covers_all_tasks():
    from exploring_meta_amd.sharding import shard_range
    for n in (1, 4, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _reduce_extras(rank, world, port):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from exploring_meta_amd.sharding import reduce_meta_batch
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    g = torch.full((5,), float(rank + 1))
    out = reduce_meta_batch(g, torch.tensor(1.0 * rank), torch.tensor(2.0), extra=[torch.tensor(3.0), 4.0])
    assert torch.equal(out[0], torch.full((5,), 3.0)) and float(out[1]) == 1.0 and float(out[2]) == 4.0
    assert [float(x) for x in out[3]] == [6.0, 8.0]
    # small tensors ride too and come back in their own shape (the BatchNorm running-statistics contribution, [2, C_total])
    out = reduce_meta_batch(g, torch.tensor(0.0), torch.tensor(0.0), extra=[1.0, torch.full((2, 3), float(rank))])
    assert float(out[3][0]) == 2.0 and out[3][1].shape == (2, 3) and torch.equal(out[3][1], torch.ones(2, 3))
    assert len(reduce_meta_batch(g, torch.tensor(1.0), torch.tensor(2.0))) == 3
    dist.destroy_process_group()


def test_validation_sums_ride_the_same_all_reduce():
    """The drivers append the validation loss / accuracy sums to the gradient's all-reduce (every rank logs the global values)."""
    mp.spawn(_reduce_extras, args=(2, _free_port()), nprocs=2, join=True)
