"""Multi-GPU layout: one process per GPU, the ensemble axis sharded across ranks.

The forward model has no coupling between members, so ranks never
exchange state.  The single collective is an all-gather of the per-rank output
statistics block -- per (variable, step, site) sum and sum of squares over the rank's
members, produced by the wavefront-shuffle reduction kernel -- from which every rank
forms the ensemble mean / variance.  On ROCm the "nccl" backend is RCCL over xGMI; the
same code runs on "gloo" CPU tensors in the tests.
"""
import numpy as np


def shard_members(n_total, world, rank):
    """Contiguous member range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sites(n_sites, world, rank):
    """Contiguous site range for configurations that shard whole sites (config C4)."""
    return shard_members(n_sites, world, rank)


def _gather0(x, world, group):
    """all_gather_into_tensor along a new leading axis (the output is the concatenation
    along dim 0, which both RCCL and gloo accept)."""
    import torch
    import torch.distributed as dist
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out.view((world,) + tuple(x.shape))


def all_gather_stats(stats, group=None):
    """stats: tensor [..., 2] (sum, sum of squares) of this rank -> [world, ..., 2].
    One all_gather_into_tensor; a single process returns stats[None]."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats.unsqueeze(0)
    world = dist.get_world_size(group)
    return _gather0(stats, world, group)


def combine_stats(gathered, counts):
    """gathered[world, ..., 2], counts[world] members per rank -> (mean, variance) of the
    whole ensemble (population variance)."""
    tot = gathered.sum(0)
    n = float(sum(counts))
    mean = tot[..., 0] / n
    var = tot[..., 1] / n - mean * mean
    return mean, var.clamp_min(0) if hasattr(var, "clamp_min") else np.maximum(var, 0)


def all_gather_planes(planes, group=None):
    """Optional full-block gather: planes[3][T][ncol_local] -> [world][3][T][ncol_local].
    Message size per rank is 3*T*ncol*elem bytes -- see DESIGN.md for why the statistics
    block, not this, is the default exchange."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return planes.unsqueeze(0)
    world = dist.get_world_size(group)
    return _gather0(planes, world, group)
