"""N>1 path on CPU: two gloo processes shard the ensemble, all-gather their statistics
blocks and agree with the single-process answer.  (On the GPU box the same functions run
over RCCL; the step kernel itself needs no collective.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sipnet_amd import dist as sd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_members_covers_everything():
    for n, w in [(10240, 8), (1000, 3), (7, 8), (65536, 2)]:
        spans = [sd.shard_members(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_members, n_steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    rng = np.random.default_rng(7)
    full = rng.standard_normal((3, n_steps, n_members))       # stands in for NEE/GPP/ET planes
    lo, hi = sd.shard_members(n_members, world, rank)
    mine = torch.from_numpy(full[:, :, lo:hi])
    stats = torch.stack([mine.sum(-1), (mine * mine).sum(-1)], dim=-1)[:, :, None, :]  # [3][T][1 site][2]
    gathered = sd.all_gather_stats(stats)
    counts = [b - a for a, b in (sd.shard_members(n_members, world, r) for r in range(world))]
    mean, var = sd.combine_stats(gathered, counts)
    planes = sd.all_gather_planes(torch.from_numpy(np.ascontiguousarray(full[:, :, lo:lo + min(counts)])))
    np.save(os.path.join(out_dir, f"mean{rank}.npy"), mean.numpy())
    np.save(os.path.join(out_dir, f"var{rank}.npy"), var.numpy())
    np.save(os.path.join(out_dir, f"planes_shape{rank}.npy"), np.array(planes.shape))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    world, n_members, n_steps = 2, 101, 37
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_members, n_steps, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(7)
    full = rng.standard_normal((3, n_steps, n_members))
    for r in range(world):
        mean = np.load(tmp_path / f"mean{r}.npy")[:, :, 0]
        var = np.load(tmp_path / f"var{r}.npy")[:, :, 0]
        assert np.allclose(mean, full.mean(-1), atol=1e-12)
        assert np.allclose(var, full.var(-1), atol=1e-12)
        assert list(np.load(tmp_path / f"planes_shape{r}.npy")) == [2, 3, n_steps, 50]
