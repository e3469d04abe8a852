"""The N > 1 path of bench.py as the driver launches it -- `python -m torch.distributed.run
--nproc-per-node 2 ... bench.py --gpus 2` -- started as a FRESH child process (the launcher runs
before anything touches the GPU), rehearsed on the one GPU of this box: both ranks share device
0 and the collectives go over gloo through host copies (`--rehearse`), everything else -- member
sharding, double-buffered planes, side-stream statistics, the all-gather of the statistics
block, the segmented full-plane gather, the barrier / max-over-ranks timing -- is the code the
8-GPU run executes.  The gathered ensemble statistics must equal a single-process run over the
same 2 x M members."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
BENCH = os.path.join(helpers.REPO, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


@pytest.mark.parametrize("workload,members,nsteps", [("c2", 192, 48 * 30), ("c4", 64, 48 * 10)])
def test_two_ranks_rehearsed_on_one_gpu_equal_a_single_process(workload, members, nsteps, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = str(tmp_path / "two.npy")
    one = str(tmp_path / "one.npy")
    common = ["--workload", workload, "--nsteps", str(nsteps), "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--no-fill-probe"]
    r2 = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2",
         "--rehearse", "--members", str(members), "--dump-stats", two] + common,
        capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    j2 = _last_json(r2.stdout)
    assert j2["n_gpus"] == 2 and j2["config"]["ranks_seen"] == 2
    assert j2["config"]["devices_seen"] == 1          # rehearsal: both ranks on device 0 (8-GPU run: 8)
    assert j2["config"]["gather_full"]["ms"] > 0 and j2["config"]["gather_full"]["segments"] == 10
    assert j2["parity"]["max_abs_dNEE"] < 1e-9
    if workload == "c2":
        # the same site, members 0 .. 2M-1 in one process
        r1 = subprocess.run([sys.executable, BENCH, "--members", str(2 * members), "--dump-stats", one] + common,
                            capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
        assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
        a, b = np.load(two), np.load(one)
        assert a.shape == b.shape == (3, nsteps, 1, 2)
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-12)
    else:
        # c4 shards whole sites: rank r owns sites 32r .. 32r+31 with members 0 .. M-1 of the ensemble
        # slice it was dealt; the combined block lists 64 sites
        a = np.load(two)
        assert a.shape == (3, nsteps, 64, 2) and np.isfinite(a).all()
        assert not np.allclose(a[:, :, 0], a[:, :, 32])          # different forcing per site


def test_bench_launches_its_own_ranks():
    """the driver's command without a launcher in front: `python3 bench.py --gpus 2 ...` starts torch.distributed.run as a
    child before anything touches the GPU and relays rank 0's one JSON line and the child's return code"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--workload", "c2", "--nsteps", "480",
                        "--steps", "2", "--warmup", "1", "--no-fill-probe", "--no-end-to-end"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["scaling"] == "weak"
    assert j["parity"]["max_abs_dNEE"] < 1e-9


@pytest.mark.parametrize("workload,members,kernel", [("c2", 1024, "stepCoopSumsKernel"), ("c3", 65536, "stepCoopSumsAtKernel<float")])
def test_two_ranks_gather_every_members_daily_sums(workload, members, kernel):
    """`--gather sums`: the timed passes all-gather every member's NEE / GPP / ET sums over 48 steps, summed inside the step
    kernel's own launch (no planes written), under the next pass; the line's gather_sums leg (the same block in four
    segments) is compared with the planes of a plain pass by bench.py itself"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--workload", workload, "--nsteps", str(48 * 20 + 7),
                        "--gather", "sums", "--steps", "2", "--warmup", "1", "--no-fill-probe", "--no-end-to-end", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = _last_json(r.stdout)
    gs = j["config"]["gather_sums"]
    assert j["config"]["gather"] == "sums" and j["config"]["ranks_seen"] == 2
    assert gs["kernel"].startswith(kernel) and gs["segments"] == 4 and gs["sum_steps"] == 48
    assert gs["bytes_sent_per_rank"] == 3 * 21 * members * 8
    assert gs["max_abs_diff_vs_planes"] < 1e-9          # (fp32-mixed: the float planes widened and added in step order -- the same bits)
    assert j["parity"]["max_abs_dNEE"] < (1e-9 if workload == "c2" else 1e-4)


@pytest.mark.parametrize("exchange", ["peer", "alltoall"])
def test_particle_filter_cycle_two_ranks_rehearsed(tmp_path, exchange):
    """c5 with two ranks (two PROCESSES) on this one GPU.  peer: likelihood weights -> ONE all-gather of the
    log-weight blocks -> each rank resamples its own particles over the gathered weights and reads every
    ancestor where it lives -- the other process's checkpoint matrices, mapped through hipIpc handles
    exchanged once (on an 8-GPU node: peer HBM over xGMI).  alltoall: all-gather of log-weights -> redundant
    resampling -> exchange plan -> ONE all-to-all of packed checkpoints.  Either way parameters travel with
    the particles and the next cycle's setupModel() runs on the RESAMPLED parameter sets.  The collectives go
    over gloo through host copies; the kernels and the bookkeeping are the 8-GPU code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2",
         "--rehearse", "--workload", "c5", "--members", "4096", "--steps", "3", "--warmup", "1",
         "--no-cpu-baseline", "--no-fill-probe", "--pf-exchange", exchange],
        capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = _last_json(r.stdout)
    pf = j["config"]["particle_filter"]
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2
    assert pf["exchange"] == exchange                      # (peer: the IPC mapping between the two processes worked)
    assert pf["received"] + pf["sent"] > 0                 # particles did cross between the ranks
    assert 1 < pf["unique_ancestors"] < 2 * 4096
    assert j["parity"]["max_abs_dNEE"] < 2e-6


# ---- the real RCCL backend, one rank -----------------------------------------------------------
# Everything of the N-rank path that does not need a second GPU, executed under "nccl" (= RCCL):
# process-group initialisation with device_id, all_gather_into_tensor on device views, the
# side-stream ordering of reductions + collective against the next pass's step kernel, the
# segmented full gather, and the particle filter's all-gather / all_to_all_single with split sizes.
def _rccl_one_rank(args, tmp_path, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "1",
         "--force-dist", "--no-cpu-baseline", "--no-fill-probe"] + args,
        capture_output=True, text=True, timeout=timeout, env=env, cwd=helpers.REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return _last_json(r.stdout)


@pytest.mark.parametrize("workload,members,nsteps", [("c2", 192, 48 * 30), ("c4", 64, 48 * 10)])
def test_rccl_one_rank_statistics_equal_the_plain_run_bit_for_bit(workload, members, nsteps, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    d, p = str(tmp_path / "dist.npy"), str(tmp_path / "plain.npy")
    common = ["--workload", workload, "--members", str(members), "--nsteps", str(nsteps), "--steps", "3",
              "--warmup", "1"]
    j = _rccl_one_rank(common + ["--dump-stats", d], tmp_path)
    assert j["config"]["dist_overhead"]["backend"] == "nccl"
    assert j["config"]["ranks_seen"] == 1 and j["config"]["devices_seen"] == 1
    assert j["config"]["gather"] == "stats"
    assert j["config"]["gather_full"]["ms"] > 0 and j["config"]["gather_full"]["segments"] == 10
    assert j["config"]["dist_overhead"]["ms_per_step_plain"] > 0
    assert j["parity"]["max_abs_dNEE"] < 1e-9
    r1 = subprocess.run([sys.executable, BENCH, "--dump-stats", p, "--no-cpu-baseline", "--no-fill-probe"] + common,
                        capture_output=True, text=True, timeout=600, env=env, cwd=helpers.REPO)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    a, b = np.load(d), np.load(p)
    assert a.shape == b.shape and np.array_equal(a, b)       # bit for bit


@pytest.mark.parametrize("exchange,collective", [("peer", "direct"), ("peer", "torch"), ("alltoall", "torch")])
def test_rccl_one_rank_particle_filter_cycle(tmp_path, exchange, collective):
    """... the peer exchange's all-gather through the engine's own RCCL communicator on the batch's stream (sipnet_comm_*,
    bench.py's default) and through torch.distributed's process group: the same ancestors either way"""
    j = _rccl_one_rank(["--workload", "c5", "--members", "4096", "--steps", "3", "--warmup", "1",
                        "--pf-exchange", exchange, "--pf-collective", collective], tmp_path)
    pf = j["config"]["particle_filter"]
    assert pf["exchange"] == exchange
    if exchange == "peer":
        assert ("sipnet_comm_all_gather" in pf["collective"]) == (collective == "direct"), pf["collective"]
    assert j["config"]["dist_overhead"]["backend"] == "nccl"
    assert pf["unique_ancestors"] > 64 and pf["ess"] > 64
    assert pf["sent"] == 0 and pf["received"] == 0          # one rank: every ancestor is local
    assert j["parity"]["max_abs_dNEE"] < 2e-6


def test_the_engines_own_communicator_one_rank():
    """sipnet_comm_*: ncclGetUniqueId / ncclCommInitRank / ncclAllGather through the C boundary, one rank (RCCL refuses two ranks
    on one device): in place and out of place, on a stream of the caller's choice, the library PyTorch already holds"""
    import ctypes as C
    import torch
    import sipnet_amd as sa
    from sipnet_amd import dist as sd
    from sipnet_amd._lib import lib
    comm = sd.DirectComm(0, 1, 0)
    assert lib().sipnet_comm_world(comm.h) == 1
    x = torch.arange(4096, dtype=torch.float64, device="cuda:0")
    out = torch.zeros((1, 4096), dtype=torch.float64, device="cuda:0")
    side = torch.cuda.Stream(device=0)
    with torch.cuda.stream(side):
        comm.all_gather(x, out, C.c_void_p(side.cuda_stream))
        y = x * 2
        comm.all_gather(y, y.view(1, 4096), C.c_void_p(side.cuda_stream))     # in place
    side.synchronize()
    assert torch.equal(out[0], x) and torch.equal(y, x * 2)
    with pytest.raises(sa.SipnetError):
        sd.DirectComm(0, 1, 99)                                                # no such device
    comm.close()
