"""Randomised differential test (tools/fuzz_gpu.py): random valid flag sets, perturbed members
with a few pushed to mortality / drought, random event schedules (all seven types, clear-cuts and
re-planting), one to eight sites, random segmentation of the run, both math policies and precisions, every kernel
(cooperative with LDS / HBM ring, one-wave, run-time flags, strict) against the oracle.  1 970
trials of it (incl. multi-site batches) ran clean when it was written (worst fp64 error 2.3e-13 of a plane's maximum); a
short fixed-seed run stays in the suite."""
import os
import subprocess
import sys

import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 11])
def test_fuzz_gpu_against_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "30", str(seed)],
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "30 trials ok" in r.stdout


REF_BIN = os.path.join(helpers.REPO, "oracle", "_ref", "sipnet_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/sipnet_ref not on this box")
def test_fuzz_cli_against_the_reference_binary():
    """tools/fuzz_cli.py: the drop-in CLI and the REAL reference binary (built in place from the
    reference's sources, travels with the repository snapshot) side by side on randomised run
    directories -- flags, perturbed parameter files, event files incl. ones the reference must
    reject, output options.  Same exit codes; sipnet.config identical; sipnet.out / events.out /
    single-variable files token for token (numbers within the last printed digit); half of the
    trials also split the run at a random midnight and hand the checkpoint over in both
    directions.  When written: 550 trials; of 1 237 output files tallied 1 227 byte-identical, the
    rest off in one to five lines; 103 two-way restart interchanges reproduced the continuous run."""
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_cli.py"), "16", "5"],
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "16 trials ok" in r.stdout


@pytest.mark.parametrize("kernel", ["coop_lds", "coop_hbm", "coop_pair", "coop_quad"])
def test_single_member_killed_by_a_harvest_then_regular_tiles(kernel):
    """the trial on which a 2 500-trial campaign found a wrong result: ONE member of a wavefront is
    killed by a harvest while its 63 neighbours live on through regular 16-step tiles.  The carbon
    wave's "ring epochs are clean" flag had been cleared in the dying lane only, so the survivors took
    the regular tiles and the dead member the general step of the same tiles afterwards, reading the
    factor slots of sixteen steps later (its NEE off by 2e-3 of the plane maximum and growing).
    The flag is wave-uniform now; this replays the trial on every cooperative layout."""
    env = dict(os.environ, FUZZ_KERNEL=kernel)
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "2500", "20261002", "1709"],
                       capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "trial 1709" in r.stdout and "forced-" + kernel in r.stdout


def test_fuzz_cooperative_layouts_with_fragile_members():
    """a fixed-seed slice of the cooperative-only campaign (FUZZ_COOP=1: default flags, throughput
    arithmetic, the one-, two- and four-chunk layouts and the default policy in turn, a few stands so
    small that single harvests finish them off in the middle of a run)"""
    env = dict(os.environ, FUZZ_COOP="1", FUZZ_BOUNDED="1")     # (the build with bounded waits: a hang would be a report)
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "40", "7"],
                       capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "40 trials ok" in r.stdout


def test_fuzz_nitrogen_cycle_cooperative_kernel():
    """a fixed-seed slice of the nitrogen-cycle campaign (FUZZ_NCYC=1: litter pool + anaerobic + nitrogen
    cycle; stepCoopNKernel forced or picked by the shape policy, the one-wave kernel of the same flag set,
    regular tiles on and off, fragile stands, productive stands short of nitrogen, random events and
    launch cuts).  Trial 127 of seed 777 is in the slice: the run on which the soil wave once read the
    light wave's factor rows of step t + 2 for step t (the slot's re-use was guarded by the carbon wave's
    progress only) -- a timing-dependent 2e-5 on NEE that the fixed tests never showed."""
    env = dict(os.environ, FUZZ_NCYC="1", FUZZ_BOUNDED="1")
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "140", "777"],
                       capture_output=True, text=True, timeout=1200, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "140 trials ok" in r.stdout and "coop-ncycle" in r.stdout


def test_fuzz_optional_physics_cooperative_kernels():
    """a fixed-seed slice of the optional-physics campaign (FUZZ_OPT=1: growth respiration, leaf water, flooding,
    litter pool, carbon saturation, anaerobic -- alone or on top of the nitrogen cycle; the cooperative layouts'
    run-time-flag instantiations forced in turn or picked by the shape policy, regular tiles on and off, fragile
    stands, random events and launch cuts)"""
    env = dict(os.environ, FUZZ_OPT="1", FUZZ_BOUNDED="1")
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "60", "41"],
                       capture_output=True, text=True, timeout=1200, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "60 trials ok" in r.stdout and "stepCoopX" in r.stdout and "stepCoopNX" in r.stdout


@pytest.mark.parametrize("campaign", ["FUZZ_COOP", "FUZZ_OPT", "FUZZ_NCYC", ""], ids=["coop", "opt", "ncyc", "all"])
def test_fuzz_sites_of_different_lengths(campaign):
    """a fixed-seed slice with FUZZ_RAGGED=1 on top of each campaign: 2-8 sites per batch whose forcings end at
    different records (one step, one step around a 16-step tile, anywhere), the launch cut before, at and after those
    ends, every kernel family; each site against the oracle run over ITS forcing"""
    env = dict(os.environ, FUZZ_RAGGED="1", FUZZ_BOUNDED="1")
    if campaign:
        env[campaign] = "1"
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), "30", "515"],
                       capture_output=True, text=True, timeout=1200, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "30 trials ok" in r.stdout


def test_fuzz_in_kernel_sums_on_the_cooperative_layouts():
    """a fixed-seed slice of the round-6 campaign (FUZZ_COOP=1 FUZZ_SUMS=1): every eligible trial (fp64, lean) is run again
    through sipnet_batch_run_sums -- random group lengths, launches cut at random multiples of the group -- and its sums must
    equal the trial's own planes added up in step order, bit for bit, and leave the same state; sites of different lengths
    in the second slice"""
    for extra, n, seed in ((dict(), "40", "9"), (dict(FUZZ_RAGGED="1"), "25", "10")):
        env = dict(os.environ, FUZZ_COOP="1", FUZZ_SUMS="1", **extra)
        r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "tools", "fuzz_gpu.py"), n, seed],
                           capture_output=True, text=True, timeout=900, env=env)
        print(r.stdout[-3000:])
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert n + " trials ok" in r.stdout and r.stdout.count(" sums(k=") >= 5
