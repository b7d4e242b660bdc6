"""Host logic of the BatchNorm running-statistics fold (core_functions/vision_models.py) against torch.nn.BatchNorm2d itself, on the
CPU: the closed-form position-weighted fold == F sequential train-mode forward passes of a real BatchNorm2d (momentum 0.1, unbiased
running variance, num_batches_tracked), and per-rank contributions add up to the single-rank fold."""
import torch

from exploring_meta_amd.core_functions.vision_models import ConvBase, apply_running_stats, running_stats_contribution
from exploring_meta_amd.engine import ModelSpec


def _passes(F, n, c, hw, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, c, hw, hw, generator=g, dtype=torch.float64) * (1.0 + 0.1 * i) + 0.05 * i for i in range(F)]


def test_fold_equals_sequential_batchnorm_updates():
    spec = ModelSpec.omniglot(5)                                     # 4 blocks x 64 filters, 28 -> 14 -> 7 -> 4 -> 2
    base = ConvBase(output_size=64, hidden=64, channels=1, max_pool=False, layers=4)
    F, n = 7, 5
    geo = [(14, 14), (7, 7), (4, 4), (2, 2)]
    bns = [torch.nn.BatchNorm2d(64, momentum=0.1).double().train() for _ in geo]
    stats = torch.zeros(F, 2, 4 * 64, dtype=torch.float64)
    for l, (bn, (h, w)) in enumerate(zip(bns, geo)):
        for i, z in enumerate(_passes(F, n, 64, h, 10 + l)):
            bn(z)                                                    # torch's own running-statistics update
            stats[i, 0, l * 64:(l + 1) * 64] = z.mean(dim=(0, 2, 3))
            stats[i, 1, l * 64:(l + 1) * 64] = z.var(dim=(0, 2, 3), unbiased=False)
    contrib = running_stats_contribution(stats.float(), torch.arange(F), F)
    apply_running_stats(base, spec, contrib, F, n)
    for blk, bn in zip(base, bns):
        assert int(blk.normalize.num_batches_tracked) == F == int(bn.num_batches_tracked)
        assert torch.allclose(blk.normalize.running_mean.double(), bn.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(blk.normalize.running_var.double(), bn.running_var, rtol=1e-5, atol=1e-6)
    # a second iteration continues from the buffers of the first
    stats2 = stats * 0.5 + 0.1
    for l, bn in enumerate(bns):
        for i in range(F):
            m, v = stats2[i, 0, l * 64:(l + 1) * 64], stats2[i, 1, l * 64:(l + 1) * 64]
            cnt = n * geo[l][0] * geo[l][1]
            bn.running_mean.mul_(0.9).add_(0.1 * m)
            bn.running_var.mul_(0.9).add_(0.1 * v * cnt / (cnt - 1))
    apply_running_stats(base, spec, running_stats_contribution(stats2.float(), torch.arange(F), F), F, n)
    for blk, bn in zip(base, bns):
        assert torch.allclose(blk.normalize.running_mean.double(), bn.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(blk.normalize.running_var.double(), bn.running_var, rtol=1e-5, atol=1e-6)
        assert int(blk.normalize.num_batches_tracked) == 2 * F


def test_contributions_of_task_shards_add_up():
    """Two ranks folding their own halves (global pass positions) == one rank folding everything: what rides in the all-reduce."""
    g = torch.Generator().manual_seed(3)
    P, T, C = 3, 5, 8
    stats = torch.rand(P, T, 2, C, generator=g)
    pos = (torch.arange(T).reshape(1, -1) * 2 + 1) * P + torch.arange(P).reshape(-1, 1)       # phase 1 of 2
    whole = running_stats_contribution(stats, pos, 2 * T * P)
    parts = running_stats_contribution(stats[:, :2], pos[:, :2], 2 * T * P) + running_stats_contribution(stats[:, 2:], pos[:, 2:], 2 * T * P)
    assert torch.allclose(whole, parts, rtol=1e-6, atol=1e-9)
    r = torch.zeros(2, C, dtype=torch.float64)                       # the recurrence it stands for
    seq = {int(pos[p, t]): stats[p, t].double() for p in range(P) for t in range(T)}
    for i in range(2 * T * P):
        r = 0.9 * r + (0.1 * seq[i] if i in seq else 0.0)
    assert torch.allclose(whole.double(), r, rtol=1e-6, atol=1e-9)
