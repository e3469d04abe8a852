"""Sanitizer builds of the host side (no GPU, no HIP: g++).  `make -C sipnet_amd/csrc san` builds
  * build/san/fuzz_host_io   host_io.cpp + restart_io.cpp + plan.cpp + ensemble_io.cpp under AddressSanitizer +
                             UndefinedBehaviourSanitizer with the mutation fuzzer tools/san/fuzz_host_io.cpp
  * build/san/shard_pool_*   the node object's host threading (csrc/shard_pool.h: persistent shard threads, task
                             hand-over, the barrier, a failing shard) under ThreadSanitizer and under ASan + UBSan
and this module runs them: every input file the reference's tests hold for the parsers (tests/golden: .param, .clim,
events.in, SIPNET_RESTART checkpoints) as a seed, a fixed number of deterministic mutations of each.  A sanitizer
finding aborts the binary; what the parsers return for garbage is not the point here (tests/test_host_io.py,
tests/test_restart_io.py pin that)."""
import glob
import gzip
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sipnet_amd", "csrc")
SAN = os.path.join(REPO, "build", "san")
GOLD = os.path.join(REPO, "tests", "golden")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


@pytest.fixture(scope="module")
def san_build():
    r = subprocess.run(["make", "-s", "-C", CSRC, "san"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return SAN


def seeds(tmp):
    out = []
    for pat in ("smoke/*/sipnet.param", "synth/allflags.param", "restart/*/run.param",
                "smoke/*/events.in", "restart/*/events_seg*.in", "events_infra/*.in",
                "restart/*/*.restart", "restart/*/*.clim", "sipnet_infra/*.clim", "sipnet_infra/*.param"):
        out += sorted(glob.glob(os.path.join(GOLD, pat)))
    for gz in sorted(glob.glob(os.path.join(GOLD, "smoke", "*", "sipnet.clim.gz"))) + \
            [os.path.join(GOLD, "synth", "halfhourly.clim.gz")]:
        dst = os.path.join(tmp, os.path.basename(os.path.dirname(gz)) + "_" + os.path.basename(gz)[:-3])
        with gzip.open(gz, "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
        out.append(dst)
    return out


def run_san(exe, *args, timeout=600):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, env=env)


def test_parsers_plan_builder_and_block_writer_under_asan_ubsan(san_build, tmp_path):
    files = seeds(str(tmp_path))
    # seeds the reference's own tests expect to FAIL are part of the corpus of mutations, not of the must-parse set:
    # the harness insists that an unmutated seed parses, so hand it the good ones only
    good = []
    for f in files:
        r = run_san(os.path.join(san_build, "fuzz_host_io"), "0", str(tmp_path), f)
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        if r.returncode == 0:
            good.append(f)
    kinds = {k: sum(1 for f in good if f.endswith(k)) for k in (".param", ".clim", ".in", ".restart")}
    assert kinds[".param"] >= 6 and kinds[".clim"] >= 6 and kinds[".in"] >= 5 and kinds[".restart"] >= 8, kinds
    r = run_san(os.path.join(san_build, "fuzz_host_io"), "120", str(tmp_path), *good)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "no sanitizer finding" in r.stdout
    n_parsed = int(r.stdout.split("fuzz_host_io:")[1].split()[0])
    assert n_parsed > len(good)          # some mutants still parse (and went through the plan builder)
    # the host's share of a device-built plan (buildSitePlanLight) was held against the full builder, resumed segments too
    tail = r.stdout.split("light plan pass held against the full builder")[1]
    assert int(tail.split()[0]) > 50 and "included: 1" in tail, tail[:120]


def test_shard_pool_under_tsan_and_asan(san_build):
    for exe in ("shard_pool_tsan", "shard_pool_asan"):
        r = run_san(os.path.join(san_build, exe))
        assert r.returncode == 0 and "shard_pool: ok" in r.stdout, exe + ": " + (r.stdout + r.stderr)[-4000:]


def test_plan_pool_under_tsan_and_asan(san_build):
    """the plan threads (csrc/plan_pool.h): jobs of every shape from one caller and from four callers at once"""
    for exe in ("plan_pool_tsan", "plan_pool_asan"):
        r = run_san(os.path.join(san_build, exe))
        assert r.returncode == 0 and "plan pool: ok" in r.stdout, exe + ": " + (r.stdout + r.stderr)[-4000:]
