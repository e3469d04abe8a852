"""The `sipnet`-compatible CLI (sipnet_amd/bin/sipnet): process-level drop-in boundary.
CPU part: option / sipnet.in precedence, --dump-config format, exit codes.
GPU part: the reference's four smoke directories reproduce their committed goldens."""
import os
import shutil
import subprocess

import pytest

from tests import helpers

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "sipnet_amd", "bin", "sipnet")


def stage(case, dst):
    d = helpers.smoke_dir(case)
    for f in ("sipnet.in", "sipnet.param", "events.in"):
        shutil.copyfile(os.path.join(d, f), os.path.join(dst, f))
    src = os.path.join(d, "sipnet.clim.gz")
    if not os.path.exists(src):
        src = os.path.join(helpers.GOLDEN, "smoke", "russell_1", "sipnet.clim.gz")
    helpers.gunzip_to(src, os.path.join(dst, "sipnet.clim"))


def run_cli(cwd, *args):
    return subprocess.run([CLI, *args], cwd=cwd, capture_output=True, text=True, timeout=300)


def config_body(text):
    return text.split("\n", 1)[1]          # first line carries the wall-clock time


@pytest.mark.parametrize("case", helpers.SMOKE_CASES)
def test_dump_config_matches_reference_golden(case, tmp_path):
    """Written before any model work, so it needs no GPU: names, sources (DEFAULT /
    INPUT_FILE / COMMAND_LINE / CALCULATED), ordering and column widths of context.c:225-268."""
    stage(case, tmp_path)
    os.remove(tmp_path / "sipnet.clim")     # stop right after the config dump (exit 6)
    r = run_cli(tmp_path, "-i", "sipnet.in")
    assert r.returncode == 6, r.stdout
    got = open(tmp_path / "sipnet.config").read()
    gold = open(os.path.join(helpers.smoke_dir(case), "sipnet.config")).read()
    assert config_body(got) == config_body(gold)


def test_cli_precedence_and_exit_codes(tmp_path):
    stage("niwot", tmp_path)
    os.remove(tmp_path / "sipnet.clim")
    # command line beats sipnet.in (cli.c / context.c:17-25)
    r = run_cli(tmp_path, "-i", "sipnet.in", "--no-print-header", "--litter-pool", "--dump-config")
    cfg = open(tmp_path / "sipnet.config").read()
    assert "LITTER_POOL  COMMAND_LINE" in cfg.replace("   ", " ").replace("  ", " ") or \
        [l for l in cfg.splitlines() if "LITTER_POOL" in l and "COMMAND_LINE" in l and l.rstrip().endswith("1")]
    # flag coupling rules -> exit 3 (context.c:195-223)
    assert run_cli(tmp_path, "--nitrogen-cycle").returncode == 3
    assert run_cli(tmp_path, "--soil-phenol").returncode == 3
    assert run_cli(tmp_path, "--no-water-hresp", "--anaerobic").returncode == 3
    # unknown option -> usage + exit 8; -v / -h -> 0
    assert run_cli(tmp_path, "--frobnicate").returncode == 8
    assert run_cli(tmp_path, "-v").returncode == 0 and "SIPNET version" in run_cli(tmp_path, "-v").stdout
    assert run_cli(tmp_path, "-h").returncode == 0
    # missing sipnet.in -> exit 6
    assert run_cli(tmp_path, "-i", "nope.in").returncode == 6
    # a parameter file without a required parameter -> exit 5
    txt = open(tmp_path / "sipnet.param").read().replace("aMaxFrac", "! aMaxFrac")
    open(tmp_path / "sipnet.param", "w").write(txt)
    assert run_cli(tmp_path, "-i", "sipnet.in").returncode == 5


@pytest.mark.gpu
@pytest.mark.parametrize("case", helpers.SMOKE_CASES)
def test_cli_reproduces_reference_smoke_goldens(case, tmp_path):
    """`sipnet -i sipnet.in` in the reference's smoke directories: sipnet.out, events.out and
    sipnet.config equal the files committed in the reference repository (russell_4, which the
    reference's smoke driver skips: the files its binary writes today)."""
    stage(case, tmp_path)
    r = run_cli(tmp_path, "-i", "sipnet.in")
    assert r.returncode == 0, r.stdout + r.stderr
    import gzip
    gold_out = gzip.open(os.path.join(helpers.smoke_dir(case), "sipnet.out.gz"), "rb").read()
    assert open(tmp_path / "sipnet.out", "rb").read() == gold_out
    if case == "russell_4":      # EVENTS = 0: the reference writes no events.out
        assert not os.path.exists(tmp_path / "events.out")
    else:
        gold_ev = open(os.path.join(helpers.smoke_dir(case), "events.out"), "rb").read()
        assert open(tmp_path / "events.out", "rb").read() == gold_ev
    got = open(tmp_path / "sipnet.config").read()
    gold = open(os.path.join(helpers.smoke_dir(case), "sipnet.config")).read()
    assert config_body(got) == config_body(gold)


@pytest.mark.gpu
def test_cli_ensemble_extension(tmp_path):
    """--ensemble-params: every row of the table is one member of ONE batch; member 0 with
    unchanged values equals the single run."""
    stage("niwot", tmp_path)
    open(tmp_path / "members.txt", "w").write("aMax psnTOpt\n8.3 24\n9.0 22.5\n7.1 25\n")
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt")
    assert r.returncode == 0, r.stdout
    import gzip
    gold_out = gzip.open(os.path.join(helpers.smoke_dir("niwot"), "sipnet.out.gz"), "rb").read()
    assert open(tmp_path / "sipnet.0.out", "rb").read() == gold_out
    a, b = open(tmp_path / "sipnet.1.out").read(), open(tmp_path / "sipnet.2.out").read()
    assert a != b and len(a.splitlines()) == 5237
