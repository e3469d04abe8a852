"""The `sipnet`-compatible CLI (sipnet_amd/bin/sipnet): process-level drop-in boundary.
CPU part: option / sipnet.in precedence, --dump-config format, exit codes.
GPU part: the reference's four smoke directories reproduce their committed goldens."""
import os
import shutil
import subprocess

import numpy as np
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
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--math", "strict")
    assert r.returncode == 0, r.stdout
    import gzip
    gold_out = gzip.open(os.path.join(helpers.smoke_dir("niwot"), "sipnet.out.gz"), "rb").read()
    assert open(tmp_path / "sipnet.0.out", "rb").read() == gold_out
    a, b = open(tmp_path / "sipnet.1.out").read(), open(tmp_path / "sipnet.2.out").read()
    assert a != b and len(a.splitlines()) == 5237
    strict = {m: open(tmp_path / f"sipnet.{m}.out").read() for m in range(3)}
    # default for an ensemble: the throughput kernels write the record (their Full instantiations);
    # at the precision `.out` prints the files agree with the strict ones (token by token, at
    # most the last printed digit apart)
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt")
    assert r.returncode == 0, r.stdout
    for m in range(3):
        fast = open(tmp_path / f"sipnet.{m}.out").read()
        la, lb = fast.split("\n"), strict[m].split("\n")
        assert len(la) == len(lb)
        differing = 0
        for x, y in zip(la, lb):
            if x == y:
                continue
            differing += 1
            for u, v in zip(x.split(), y.split()):
                if u != v:
                    dec = len(v.split(".")[1]) if "." in v else 0
                    assert abs(float(u) - float(v)) <= 1.01 * 10 ** (-dec), (x, y)
        assert differing <= 5, differing


def test_cli_devices_list_is_validated(tmp_path):
    """--devices: malformed lists are a CLI error (exit 8) before anything else happens"""
    stage("niwot", tmp_path)
    for bad in ("x", "3-1", "1-", "-1", ",", "0-x"):
        r = run_cli(tmp_path, "-i", "sipnet.in", "--devices", bad)
        assert r.returncode == 8, (bad, r.returncode, r.stdout)


@pytest.mark.gpu
def test_cli_ensemble_sharded_over_devices_equals_one_batch(tmp_path):
    """--devices: the ensemble axis shards across HIP devices behind the C boundary (one host thread
    + one sipnet_batch per device, every shard writes its members' files).  `--devices 0,0,0` on this
    one-GPU box = three shards of a 7-member ensemble: every member's .out / events.out /
    restart checkpoint equals the single-batch run's byte for byte."""
    import filecmp
    rows = ["aMax psnTOpt soilWHC"] + ["%.3f %.2f %.2f" % (7.5 + 0.3 * i, 22.0 + 0.5 * i, 10.0 + i) for i in range(7)]
    outs = {}
    for tag, dev in (("one", "0"), ("three", "0,0,0")):
        d = tmp_path / tag
        d.mkdir()
        stage("russell_1", d)
        open(d / "members.txt", "w").write("\n".join(rows) + "\n")
        r = run_cli(d, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--devices", dev,
                    "--restart-out", "ck")           # an ensemble: throughput kernels, full records
        assert r.returncode == 0, r.stdout + r.stderr
        if tag == "three":
            assert "sharded over 3 device(s)" in r.stdout
        outs[tag] = d
    for m in range(7):
        for name in (f"sipnet.{m}.out", f"events.{m}.out"):
            assert filecmp.cmp(outs["one"] / name, outs["three"] / name, shallow=False), name
        a = [l for l in open(outs["one"] / f"ck.{m}") if "checkpoint_utc_epoch" not in l]
        b = [l for l in open(outs["three"] / f"ck.{m}") if "checkpoint_utc_epoch" not in l]
        assert a == b
    # a device that does not exist is refused up front
    r = run_cli(outs["one"], "-i", "sipnet.in", "--devices", "0,99")
    assert r.returncode == 1 and "only" in r.stdout


def test_cli_ensemble_stats_needs_an_ensemble(tmp_path):
    stage("niwot", tmp_path)
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-stats", "stats.txt")
    assert r.returncode == 8 and "--ensemble-params" in r.stdout


@pytest.mark.gpu
def test_cli_ensemble_stats_through_the_node_object(tmp_path):
    """--ensemble-stats: the ensemble runs as ONE sipnet_node (the C host object: member shards, one
    RCCL rank per device, one all-gather of the statistics block) and the CLI prints per-step mean and
    sd of NEE / GPP / ET -- equal to what the members' own .out files give (russell_1: events)."""
    M = 9
    rows = ["aMax psnTOpt soilWHC"] + ["%.3f %.2f %.2f" % (7.5 + 0.3 * i, 22.0 + 0.5 * i, 10.0 + i) for i in range(M)]
    stage("russell_1", tmp_path)
    open(tmp_path / "members.txt", "w").write("\n".join(rows) + "\n")
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-stats", "stats.txt",
                "--print-header")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL" in r.stdout
    lines = open(tmp_path / "stats.txt").read().splitlines()
    assert lines[0].split() == "year day time n meanNEE sdNEE meanGPP sdGPP meanET sdET".split()
    st = np.array([[float(x) for x in l.split()] for l in lines[1:]])
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--print-header")
    assert r.returncode == 0, r.stdout + r.stderr
    header = open(tmp_path / "sipnet.0.out").readline().split()
    assert header[0] == "year"
    outs = np.stack([np.loadtxt(tmp_path / f"sipnet.{m}.out", skiprows=1) for m in range(M)])     # [M][T][cols]
    assert st.shape[0] == outs.shape[1] and (st[:, 3] == M).all()
    for name, col, digits in (("nee", 4, 3), ("gpp", 6, 3), ("evapotranspiration", 8, 5)):
        x = outs[:, :, header.index(name)]
        # the members' files print `digits` decimals (sipnet.c:453-473): their mean is within half a unit
        np.testing.assert_allclose(st[:, col], x.mean(0), atol=0.51 * 10 ** -digits)
        np.testing.assert_allclose(st[:, col + 1], x.std(0), atol=1.1 * 10 ** -digits)
    assert st[:, 6].max() > 0.05 and st[:, 5].max() > 1e-3      # there is a signal and a spread


def test_cli_sites_option_errors_need_no_gpu(tmp_path):
    """--sites: a missing list -> 6, an empty one -> 5, combined with --ensemble-params -> 8, a run directory that
    does not exist -> 6"""
    assert run_cli(tmp_path, "--sites", "nope.txt").returncode == 6
    open(tmp_path / "runs.txt", "w").write("# nothing\n\n")
    assert run_cli(tmp_path, "--sites", "runs.txt").returncode == 5
    open(tmp_path / "runs.txt", "w").write("no_such_dir\n")
    open(tmp_path / "m.txt", "w").write("aMax\n8\n")
    assert run_cli(tmp_path, "--sites", "runs.txt", "--ensemble-params", "m.txt").returncode == 8
    assert run_cli(tmp_path, "--sites", "runs.txt").returncode == 6


@pytest.mark.gpu
@pytest.mark.parametrize("math,devices", [("auto", "0"), ("fast", "0"), ("auto", "0,0,0")], ids=["auto", "fast", "auto-3-shards"])
def test_cli_sites_stacks_run_directories_into_shared_batches(tmp_path, math, devices):
    """`sipnet --sites runs.txt`: the reference's five smoke directories (three flag sets, two step counts) plus three
    re-parameterised copies of russell_1 (same forcing: members of ONE site) in one process.  Every directory gets
    the files its own run writes: with --math auto (strict kernel) byte-identical to the reference's goldens and to
    separate runs; with --math fast (throughput kernels) to the last printed digit; with --devices naming several
    shards the same files"""
    import gzip
    dirs = []
    for case in helpers.SMOKE_CASES:
        d = tmp_path / case
        d.mkdir()
        stage(case, d)
        dirs.append(case)
    for k, amax in enumerate((6.9, 7.7, 8.8)):           # an ensemble at russell_1's site: same clim + events, other aMax
        d = tmp_path / f"ens{k}"
        d.mkdir()
        stage("russell_1", d)
        txt = open(d / "sipnet.param").read().splitlines()
        txt = [(f"aMax {amax}" if l.split() and l.split()[0] == "aMax" else l) for l in txt]
        open(d / "sipnet.param", "w").write("\n".join(txt) + "\n")
        dirs.append(f"ens{k}")
    open(tmp_path / "runs.txt", "w").write("# the smoke cases and an ensemble\n" + "\n".join(dirs) + "\n")
    # (options apply to every directory; --devices deals the sites of a flag set to the devices, one batch and one
    # host thread each -- here three shards on device 0)
    r = run_cli(tmp_path, "--sites", "runs.txt", "-i", "sipnet.in", "--math", math, "--devices", devices)
    assert r.returncode == 0, r.stdout + r.stderr
    if devices != "0":
        assert r.stdout.count("(device 0)") > 3, r.stdout
    # russell_1 + its three copies share one site (4 members); russell_1/4-like flag sets and niwot's step count differ
    assert "8 run(s) in" in r.stdout and "4 member(s)" in r.stdout, r.stdout

    def same(a, b):
        if math == "auto":
            return a == b
        ta, tb = a.split(), b.split()
        return len(ta) == len(tb) and all(x == y or abs(float(x) - float(y)) <= 1.01 * 10 ** -(len(x.split(b".")[-1]) if b"." in x else 0)
                                          for x, y in zip(ta, tb))
    for case in helpers.SMOKE_CASES:
        gold_out = gzip.open(os.path.join(helpers.smoke_dir(case), "sipnet.out.gz"), "rb").read()
        assert same(open(tmp_path / case / "sipnet.out", "rb").read(), gold_out), case
        if case == "russell_4":
            assert not os.path.exists(tmp_path / case / "events.out")
        else:
            gold_ev = open(os.path.join(helpers.smoke_dir(case), "events.out"), "rb").read()
            assert same(open(tmp_path / case / "events.out", "rb").read(), gold_ev), case
        got = open(tmp_path / case / "sipnet.config").read()
        assert config_body(got) == config_body(open(os.path.join(helpers.smoke_dir(case), "sipnet.config")).read())
    # the re-parameterised members against separate runs of the same CLI in copies of their directories
    for k in range(3):
        solo = tmp_path / f"solo{k}"
        shutil.copytree(tmp_path / f"ens{k}", solo, ignore=shutil.ignore_patterns("*.out", "*.config"))
        r1 = run_cli(solo, "-i", "sipnet.in", "--math", "strict" if math == "auto" else "fast")
        assert r1.returncode == 0, r1.stdout
        for f in ("sipnet.out", "events.out"):
            assert same(open(tmp_path / f"ens{k}" / f, "rb").read(), open(solo / f, "rb").read()), (k, f)
    assert open(tmp_path / "ens0" / "sipnet.out", "rb").read() != open(tmp_path / "ens2" / "sipnet.out", "rb").read()


def test_cli_ensemble_out_option_errors_need_no_gpu(tmp_path):
    """--ensemble-out without an ensemble, its satellites without it, an unknown column -> 8"""
    stage("niwot", tmp_path)
    assert run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-out", "e.nc").returncode == 8
    open(tmp_path / "m.txt", "w").write("aMax\n8\n")
    assert run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "m.txt", "--ensemble-out-f32").returncode == 8
    assert run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "m.txt", "--ensemble-text").returncode == 8
    assert run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "m.txt", "--ensemble-out", "e.nc",
                   "--ensemble-out-columns", "nee,frobnication").returncode == 8
    assert run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "m.txt", "--ensemble-out", "e.nc",
                   "--ensemble-stats", "s.txt").returncode == 8


def _out_table(text):
    """`.out` text -> (header names, list of rows of tokens)"""
    import sipnet_amd as sa
    lines = text.splitlines()
    if lines[0].split()[0] == "year":
        return lines[0].split(), [l.split() for l in lines[1:]]
    return sa.format_out_header().split(), [l.split() for l in lines]     # (PRINT_HEADER = 0, e.g. niwot's sipnet.in)


def _assert_block_matches_text(block, names, rows, member, where):
    """every variable of the block against the printed `.out` columns, at print precision"""
    for j, name in enumerate(names):
        if name in ("year", "day", "time"):
            continue
        col = block[name][:, member]
        for t in (0, 1, len(rows) // 2, len(rows) - 1):
            text = rows[t][j]
            digits = len(text.split(".")[1]) if "." in text else 0
            assert abs(float(text) - col[t]) <= 0.5000001 * 10 ** -digits, (where, name, t, text, col[t])


@pytest.mark.gpu
def test_cli_ensemble_out_block_against_the_goldens_and_the_members_text(tmp_path):
    """`sipnet --ensemble-params T --ensemble-out e.nc --ensemble-out-columns all --ensemble-text` on russell_1 with a
    3-member table: every variable of the block equals, at print precision, the columns of the reference's golden
    `.out` (member 0 = the unchanged parameters) and of the members' own text files; the default block (three planes
    from the lean throughput kernels, no text) and the sharded run (--devices 0,0: two shards streaming their member
    ranges into the one file) hold the same numbers; --ensemble-out-f32 halves the file"""
    import gzip
    from sipnet_amd import ensemble_io as eio
    stage("russell_1", tmp_path)
    open(tmp_path / "members.txt", "w").write("aMax psnTOpt\n" + "\n".join(
        f"{a} {o}" for a, o in ((_param(tmp_path, "aMax"), _param(tmp_path, "psnTOpt")), (8.1, 23.0), (6.4, 25.5))) + "\n")
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "all.nc",
                "--ensemble-out-columns", "all", "--ensemble-text", "--math", "strict")
    assert r.returncode == 0, r.stdout + r.stderr
    block = eio.read_ensemble_netcdf(tmp_path / "all.nc")
    _, dims, gatts, units = eio.open_ensemble_netcdf(tmp_path / "all.nc")
    gold = gzip.open(os.path.join(helpers.smoke_dir("russell_1"), "sipnet.out.gz"), "rt").read()
    names, rows = _out_table(gold)
    assert dims == {"time": len(rows), "member": 3} and gatts["math"] == "strict"
    assert [n for n in names if n not in ("year", "day", "time")] == [k for k in block if k not in ("year", "day", "hour", "length", "member")]
    _assert_block_matches_text(block, names, rows, 0, "golden")
    assert [int(r_[0]) for r_ in rows] == block["year"].tolist() and [int(r_[1]) for r_ in rows] == block["day"].tolist()
    assert open(tmp_path / "sipnet.0.out").read() == gold            # --ensemble-text: the text files as before
    for m in range(3):
        n2, rows_m = _out_table(open(tmp_path / f"sipnet.{m}.out").read())
        _assert_block_matches_text(block, n2, rows_m, m, f"member {m}")
    assert np.abs(block["nee"][:, 1] - block["nee"][:, 2]).max() > 1e-3
    assert units["soilWater"] == "cm" and units["nee"] == "g C m-2 step-1"
    # the default block: planes only, throughput kernels, no text files
    for f in os.listdir(tmp_path):
        if f.endswith(".out"):
            os.remove(tmp_path / f)
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "planes.nc")
    assert r.returncode == 0, r.stdout + r.stderr
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".out")]
    planes = eio.read_ensemble_netcdf(tmp_path / "planes.nc")
    assert set(planes) == {"year", "day", "hour", "length", "member", "nee", "gpp", "evapotranspiration"}
    for k in ("nee", "gpp", "evapotranspiration"):
        np.testing.assert_allclose(planes[k], block[k], rtol=0, atol=1e-11)
    # two shards on one device stream into ONE file; single precision
    # (... and the record held on the device 1 000 steps at a time: six launches per shard, the last one ragged)
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "sharded.nc",
                "--devices", "0,0", "--ensemble-out-f32", "--ensemble-out-columns", "nee,plantWoodC,soilWater",
                "--ensemble-out-segment", "1000")
    assert r.returncode == 0, r.stdout + r.stderr
    sh = eio.read_ensemble_netcdf(tmp_path / "sharded.nc")
    assert sh["nee"].dtype == np.float32 and sh["member"].tolist() == [0, 1, 2]
    for k in ("nee", "plantWoodC", "soilWater"):
        np.testing.assert_allclose(sh[k], block[k].astype(np.float32), rtol=2e-7, atol=1e-11)
    assert os.path.getsize(tmp_path / "sharded.nc") < 0.2 * os.path.getsize(tmp_path / "all.nc")


@pytest.mark.gpu
def test_cli_ensemble_out_sums_block(tmp_path):
    """`sipnet --ensemble-params T --ensemble-out e.nc --ensemble-out-sums 8` on niwot-like default physics (russell_4's flags are
    data): every member's sums over groups of 8 steps (russell's records are 0.125 d: daily sums), the last group shorter --
    equal, bit for bit, to the full block's planes added up in step order (throughput kernels: the sums come out of the step
    kernel's own launch; --math strict: from the planes, on the host; two shards stream into ONE file); the block's time axis
    holds each group's first record, its step length the group's; option errors are CLI errors"""
    from sipnet_amd import ensemble_io as eio
    stage("russell_4", tmp_path)                        # (no events, no GDD, soil phenology: the default-physics kernels)
    open(tmp_path / "members.txt", "w").write("aMax psnTOpt\n" + "\n".join(f"{a} {o}" for a, o in ((7.9, 24.0), (8.1, 23.0), (6.4, 25.5))) + "\n")
    K = 8
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "planes.nc")
    assert r.returncode == 0, r.stdout + r.stderr
    planes = eio.read_ensemble_netcdf(tmp_path / "planes.nc")
    T = planes["nee"].shape[0]
    G = (T + K - 1) // K
    want = {}
    for k in ("nee", "gpp", "evapotranspiration"):
        acc = np.zeros((G, 3))
        for t in range(T):
            acc[t // K] += planes[k][t]
        want[k] = acc
    for args, how in ((("--devices", "0"), "from the step kernel's launch"), (("--devices", "0,0"), "from the step kernel's launch"),
                      (("--math", "strict"), "from the planes, on the host")):
        out = "sums_%s.nc" % "_".join(a.strip("-").replace(",", "") for a in args)
        r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", out, "--ensemble-out-sums", str(K), *args)
        assert r.returncode == 0, r.stdout + r.stderr
        assert how in r.stdout, r.stdout
        blk = eio.read_ensemble_netcdf(tmp_path / out)
        _, dims, gatts, _ = eio.open_ensemble_netcdf(tmp_path / out)
        assert dims == {"time": G, "member": 3} and gatts["sums_over_steps"] == str(K)
        assert blk["year"].tolist() == planes["year"][::K].tolist() and blk["day"].tolist() == planes["day"][::K].tolist()
        np.testing.assert_allclose(blk["length"][:-1], np.add.reduceat(planes["length"], np.arange(0, T, K))[:-1], rtol=1e-12)
        for k in want:
            if "strict" in args:
                np.testing.assert_allclose(blk[k], want[k], rtol=0, atol=1e-10)       # (the strict kernel's planes, summed)
            else:
                np.testing.assert_array_equal(blk[k], want[k])
        assert os.path.getsize(tmp_path / out) < 0.2 * os.path.getsize(tmp_path / "planes.nc")
    for bad in (("--ensemble-out-sums", "0"), ("--ensemble-out-sums", "8", "--ensemble-out-columns", "nee"), ("--ensemble-out-sums", "8", "--ensemble-text")):
        r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "x.nc", *bad)
        assert r.returncode == 8, (bad, r.stdout)


def _param(d, name):
    for l in open(os.path.join(d, "sipnet.param")):
        t = l.split()
        if t and t[0] == name:
            return float(t[1])
    raise KeyError(name)


@pytest.mark.gpu
def test_cli_sites_ensemble_out_one_block_per_forcing(tmp_path):
    """--sites + --ensemble-out: one block per distinct forcing, named after the position of the site's first run;
    members = positions in the list; the numbers are those of the directories' own text files"""
    from sipnet_amd import ensemble_io as eio
    dirs = []
    for k, (case, amax) in enumerate((("russell_1", None), ("niwot", None), ("russell_1", 7.3))):
        d = tmp_path / f"r{k}"
        d.mkdir()
        stage(case, d)
        if amax is not None:
            txt = [(f"aMax {amax}" if l.split() and l.split()[0] == "aMax" else l) for l in open(d / "sipnet.param").read().splitlines()]
            open(d / "sipnet.param", "w").write("\n".join(txt) + "\n")
        dirs.append(f"r{k}")
    open(tmp_path / "runs.txt", "w").write("\n".join(dirs) + "\n")
    r = run_cli(tmp_path, "--sites", "runs.txt", "-i", "sipnet.in", "--ensemble-out", "out/blk.nc", "--ensemble-out-columns", "all",
                "--ensemble-text", "--math", "strict")
    assert r.returncode == 6                      # the directory of the block does not exist: a file-open failure
    os.mkdir(tmp_path / "out")
    r = run_cli(tmp_path, "--sites", "runs.txt", "-i", "sipnet.in", "--ensemble-out", "out/blk.nc", "--ensemble-out-columns", "all",
                "--ensemble-text", "--math", "strict")
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(tmp_path / "out")) == ["blk.0.nc", "blk.1.nc"]
    b0 = eio.read_ensemble_netcdf(tmp_path / "out" / "blk.0.nc")      # russell_1's forcing: runs 0 and 2
    b1 = eio.read_ensemble_netcdf(tmp_path / "out" / "blk.1.nc")      # niwot: run 1
    assert b0["member"].tolist() == [0, 2] and b1["member"].tolist() == [1]
    _, _, gatts, _ = eio.open_ensemble_netcdf(tmp_path / "out" / "blk.0.nc")
    assert gatts["run_dirs"].split() == [str(tmp_path / "r0"), str(tmp_path / "r2")]
    for blk, run, m in ((b0, 0, 0), (b0, 2, 1), (b1, 1, 0)):
        names, rows = _out_table(open(tmp_path / f"r{run}" / "sipnet.out").read())
        assert blk["nee"].shape[0] == len(rows)
        _assert_block_matches_text(blk, names, rows, m, f"run {run}")


@pytest.mark.gpu
def test_cli_bounded_waits_option_gives_the_same_files(tmp_path):
    """--bounded-waits: the ensemble runs on the cooperative kernels' bounded-wait build (a wait that never ends would be
    reported with exit 7 instead of hanging the GPU); same bits, hence the same text"""
    stage("niwot", tmp_path)
    open(tmp_path / "members.txt", "w").write("aMax psnTOpt\n8.3 24\n9.0 22.5\n7.1 25\n")
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "a.nc")
    assert r.returncode == 0, r.stdout + r.stderr
    r = run_cli(tmp_path, "-i", "sipnet.in", "--ensemble-params", "members.txt", "--ensemble-out", "b.nc", "--bounded-waits")
    assert r.returncode == 0, r.stdout + r.stderr
    from sipnet_amd import ensemble_io as eio
    a, b = eio.read_ensemble_netcdf(tmp_path / "a.nc"), eio.read_ensemble_netcdf(tmp_path / "b.nc")
    for k in ("nee", "gpp", "evapotranspiration"):
        np.testing.assert_array_equal(a[k], b[k])
