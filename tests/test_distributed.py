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


@pytest.mark.timeout(180)
def test_bench_starts_its_own_ranks_when_no_launcher_is_in_front():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py is its own launcher (a child
    `python -m torch.distributed.run`, bench.launch_ranks) instead of exiting 2 -- the driver's multi-GPU command may
    be given either way.  --launch-probe: the ranks only meet over gloo (no GPU here); the same command with the real
    workload runs under `-m gpu` (tests/test_gpu_multirank.py)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-probe"],
                       capture_output=True, text=True, timeout=170, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0's ONE line, relayed
    assert lines[0]["ranks_seen"] == 2 and lines[0]["processes_seen"] == 2 and lines[0]["n_gpus"] == 2
    # ... and a failing rank's return code is the command's
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "c2", "--nsteps", "48",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=170,
                         env=dict(env, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""), cwd=REPO)
    assert bad.returncode not in (0, 2), bad.stderr[-1500:]   # (2 was "needs torch.distributed.run")
