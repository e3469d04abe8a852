"""include/sipnet_amd.h consumed from plain C (gcc -std=c99 -pedantic, no C++): the header is valid C,
the library links from C, the host-side entry points work; without a GPU batch creation answers
SIPNET_ERR_NO_DEVICE (there is no CPU path), with one a two-member batch runs through the C-ABI
alone (no Python binding, no CLI) and matches the oracle."""
import os
import subprocess

import numpy as np
import pytest

import sipnet_amd as sa
from tests import helpers

SRC = os.path.join(helpers.REPO, "tests", "c", "capi_consumer.c")
NODE_SRC = os.path.join(helpers.REPO, "tests", "c", "node_consumer.c")


def build(tmp_path, src=SRC):
    exe = str(tmp_path / os.path.basename(src)[:-2])
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        "-I" + os.path.join(helpers.REPO, "include"), src, "-o", exe,
                        "-L" + os.path.join(helpers.REPO, "sipnet_amd"), "-lsipnet_amd",
                        "-Wl,-rpath," + os.path.join(helpers.REPO, "sipnet_amd")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run(exe, tmp_path):
    clim = str(tmp_path / "niwot.clim")
    helpers.gunzip_to(os.path.join(helpers.smoke_dir("niwot"), "sipnet.clim.gz"), clim)
    r = subprocess.run([exe, os.path.join(helpers.smoke_dir("niwot"), "sipnet.param"), clim],
                       capture_output=True, text=True, timeout=300)
    kv = dict(l.split("=", 1) for l in r.stdout.strip().split("\n") if "=" in l)
    return r.returncode, kv


@pytest.mark.skipif(sa.lib().sipnet_device_count() > 0, reason="a GPU is present")
def test_header_is_plain_c_and_there_is_no_cpu_path(tmp_path):
    rc, kv = run(build(tmp_path), tmp_path)
    assert rc == 0
    assert kv["n_steps"] == "5237" and kv["index_aMax"] == "4" and float(kv["aMax"]) == 8.3
    assert int(kv["header_bytes"]) > 300
    assert kv["create"] == "100" and "no usable HIP device" in kv["no_device_message"]


@pytest.mark.gpu
def test_two_member_batch_through_the_c_abi_alone(oracle, tmp_path):
    rc, kv = run(build(tmp_path), tmp_path)
    assert rc == 0 and kv["create"] == "0"
    case = helpers.load_smoke_case("niwot", str(tmp_path))
    members = np.stack([case["params"], case["params"]])
    members[1, sa.config.param_index("aMax")] *= 1.1
    want, _, st = oracle.run_block(case["flags"], members, case["clim"], None)
    assert (st == 0).all()
    assert float(kv["sum_nee_0"]) == pytest.approx(want[0][:, 0].sum(), abs=1e-8)
    assert float(kv["sum_nee_1"]) == pytest.approx(want[0][:, 1].sum(), abs=1e-8)
    assert abs(float(kv["sum_nee_0"]) - float(kv["sum_nee_1"])) > 1e-3


def run_node(exe, tmp_path, members, devices="0"):
    clim = str(tmp_path / "niwot.clim")
    helpers.gunzip_to(os.path.join(helpers.smoke_dir("niwot"), "sipnet.clim.gz"), clim)
    r = subprocess.run([exe, os.path.join(helpers.smoke_dir("niwot"), "sipnet.param"), clim, str(members), devices],
                       capture_output=True, text=True, timeout=600)
    kv = dict(l.split("=", 1) for l in r.stdout.strip().split("\n") if "=" in l)
    return r.returncode, kv, r.stdout + r.stderr


@pytest.mark.skipif(sa.lib().sipnet_device_count() > 0, reason="a GPU is present")
def test_node_object_links_from_c_and_needs_a_device(tmp_path):
    """the multi-GPU host object (sipnet_node_*, RCCL behind it) compiles and links from plain C;
    without a device it answers SIPNET_ERR_NO_DEVICE before it touches RCCL"""
    rc, kv, out = run_node(build(tmp_path, NODE_SRC), tmp_path, 130)
    assert rc == 0, out
    assert kv["create"] == "100" and "no usable HIP device" in kv["no_device_message"]


@pytest.mark.gpu
def test_node_of_one_device_all_gathers_through_rccl_from_c(oracle, tmp_path):
    """sipnet_node with devices = {0}: 130 members (a ragged third chunk) on the throughput kernel,
    sipnet_node_gather_stats and sipnet_node_gather_planes through a one-rank RCCL communicator, from
    a C99 program; the statistics equal the sums over the gathered planes and the members the oracle"""
    rc, kv, out = run_node(build(tmp_path, NODE_SRC), tmp_path, 130)
    assert rc == 0 and kv["create"] == "0", out
    assert "RCCL" in kv["collective_library"] and kv["n_devices"] == "1"
    assert kv["gathered_stats_identical"] == "1"
    assert kv["gathering_segments_equal_gathered_planes"] == "1"      # sipnet_node_run_gathering: the overlapped plane gather from C
    # sipnet_node_run_gathering_reduced from C: every member's sums over groups of 48 steps = the planes summed in step order
    assert kv["reduced_sums_equal_planes"] == "1" and float(kv["reduced_sums_max_abs"]) == 0.0, out
    assert int(kv["reduced_groups"]) == (5237 + 47) // 48 and int(kv["reduced_bytes_per_rank"]) == 3 * 110 * 130 * 8
    assert float(kv["stats_vs_planes_max_abs"]) < 1e-9
    assert kv["kernel"].startswith("stepCoopKernel<double")
    case = helpers.load_smoke_case("niwot", str(tmp_path))
    members = np.stack([case["params"], case["params"]])
    members[1, sa.config.param_index("aMax")] *= 1.0 + 0.001 * 129
    want, _, st = oracle.run_block(case["flags"], members, case["clim"], None)
    assert (st == 0).all()
    assert float(kv["sum_nee_member_0"]) == pytest.approx(want[0][:, 0].sum(), abs=1e-8)
    assert float(kv["sum_nee_member_last"]) == pytest.approx(want[0][:, 1].sum(), abs=1e-8)
    assert float(kv["sum_nee_day0_member_0"]) == pytest.approx(want[0][:48, 0].sum(), abs=1e-9)      # ... and the oracle's


RANK_SRC = os.path.join(helpers.REPO, "tests", "c", "rank_consumer.c")


def run_rank(exe, tmp_path, members, world=1, rank=0, device=0):
    clim = str(tmp_path / "niwot.clim")
    helpers.gunzip_to(os.path.join(helpers.smoke_dir("niwot"), "sipnet.clim.gz"), clim)
    r = subprocess.run([exe, os.path.join(helpers.smoke_dir("niwot"), "sipnet.param"), clim, str(members), str(world), str(rank),
                        str(device), str(tmp_path / "comm.id")], capture_output=True, text=True, timeout=600)
    kv = dict(l.split("=", 1) for l in r.stdout.strip().split("\n") if "=" in l)
    return r.returncode, kv, r.stdout + r.stderr


@pytest.mark.skipif(sa.lib().sipnet_device_count() > 0, reason="a GPU is present")
def test_rank_consumer_links_from_c_and_needs_a_device(tmp_path):
    """a process-per-GPU host (sipnet_batch_* + sipnet_comm_*) compiles as pedantic C99 and links; without a device it says so"""
    rc, kv, out = run_rank(build(tmp_path, RANK_SRC), tmp_path, 130)
    assert rc == 0, out
    assert kv["create"] == "100" and "no usable HIP device" in kv["no_device_message"]


@pytest.mark.gpu
def test_a_process_per_gpu_host_sums_in_the_launch_and_all_gathers_through_the_engines_communicator(oracle, tmp_path):
    """tests/c/rank_consumer.c with world = 1 (RCCL refuses two ranks on one device): the batch API alone, every member's daily
    sums out of the step kernel's launch (sipnet_batch_run_sums), ONE all-gather through sipnet_comm_* on the batch's stream --
    the gathered block against the oracle's daily sums"""
    rc, kv, out = run_rank(build(tmp_path, RANK_SRC), tmp_path, 130)
    assert rc == 0 and kv["create"] == "0" and kv["rc"] == "0", out
    assert kv["comm_world"] == "1" and kv["sums_in_kernel"] == "1" and kv["kernel"].startswith("stepCoopSumsKernel<"), out
    assert int(kv["groups"]) == (5237 + 47) // 48 and int(kv["bytes_per_rank"]) == 3 * 110 * 130 * 8
    case = helpers.load_smoke_case("niwot", str(tmp_path))
    members = np.stack([case["params"], case["params"]])
    members[1, sa.config.param_index("aMax")] *= 1.0 + 0.001 * 129
    want, _, st = oracle.run_block(case["flags"], members, case["clim"], None)
    assert (st == 0).all()
    assert float(kv["sum_nee_member_0"]) == pytest.approx(want[0][:, 0].sum(), abs=1e-8)
    assert float(kv["sum_nee_member_last"]) == pytest.approx(want[0][:, 1].sum(), abs=1e-8)
    assert float(kv["sum_nee_day0_member_0"]) == pytest.approx(want[0][:48, 0].sum(), abs=1e-9)


PF_SRC = os.path.join(helpers.REPO, "tests", "c", "pf_consumer.c")


def run_pf(exe, tmp_path, n, devices, cycles=5, n_steps=48, one_wave=0):
    from sipnet_amd import synth
    clim = str(tmp_path / "day.clim")
    synth.write_clim(clim, synth.round_like_file(synth.half_hourly_year_raw(n_steps)))
    r = subprocess.run([exe, os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param"), clim, str(n), devices,
                        str(cycles), str(n_steps), str(one_wave)], capture_output=True, text=True, timeout=600)
    kv = dict(l.split("=", 1) for l in r.stdout.strip().split("\n") if "=" in l)
    return r.returncode, kv, r.stdout + r.stderr


@pytest.mark.skipif(sa.lib().sipnet_device_count() > 0, reason="a GPU is present")
def test_filter_consumer_links_from_c_and_needs_a_device(tmp_path):
    rc, kv, out = run_pf(build(tmp_path, PF_SRC), tmp_path, 512, "0")
    assert rc == 0, out
    assert kv["create"] == "100" and "no usable HIP device" in kv["no_device_message"]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0", "0,0"], ids=["rccl-one-rank", "two-shards-one-gpu"])
def test_filter_cycles_through_the_node_object_from_c(tmp_path, devices):
    """config 5's cycle from a C99 program: setupModel -> forecast -> sipnet_node_pf_analysis (ONE all-gather of
    the log-weight blocks through a one-rank RCCL communicator, or between two shards of one GPU; peer-read
    resampling), 8 cycles with no host synchronisation, against the same cycles on ONE batch: identical state"""
    rc, kv, out = run_pf(build(tmp_path, PF_SRC), tmp_path, 16384, devices, cycles=8)
    assert rc == 0 and kv["create"] == "0", out
    assert kv["collective_library"].startswith("librccl" if devices == "0" else "event-ordered")
    assert kv["state_identical"] == "1" and float(kv["state_max_rel_diff"]) == 0.0, out
    assert kv["kernel"] == kv["kernel_twin"], out         # (these shapes take the same kernel; see pf_consumer.c's header)
    assert kv["params_by_index"] == "1" and kv["analysis_one_launch"] == "1"
    if devices == "0,0":
        assert float(kv["crossing_per_cycle"]) > 0, out  # particles crossed between the shards
    assert int(kv["distinct_neighbours"]) > 100          # the filter kept many distinct particles
    assert float(kv["ms_per_cycle_node"]) > 0 and float(kv["ms_per_cycle_plain"]) > 0


@pytest.mark.gpu
def test_filter_consumer_shards_and_twin_on_one_kernel_are_bit_identical(tmp_path):
    """two shards of 32 768 particles and the twin's 65 536 would take different kernels by shape (equal to rounding, not to
    bits); with the seventh argument both run the one-wave kernel: identical states"""
    rc, kv, out = run_pf(build(tmp_path, PF_SRC), tmp_path, 65536, "0,0", cycles=4, one_wave=1)
    assert rc == 0 and kv["create"] == "0", out
    assert kv["kernel"].startswith("stepFastKernel") and kv["kernel"] == kv["kernel_twin"], out
    assert kv["state_identical"] == "1" and float(kv["state_max_rel_diff"]) == 0.0, out
