"""Restart checkpoints end to end on the GPU: `--restart-out` / `--restart-in`
of the CLI and export_restart / import_restart of the batch API, pinned against checkpoints and
resumed outputs of the REAL reference (tests/golden/restart/*, made by tools/make_golden.py)
and -- where oracle/_ref/sipnet_ref travelled to this box -- against the reference binary
itself resuming from OUR checkpoint.

Cases: the reference's own restart test (testRestartMVP.c data: fertiliser + tillage in
segment 1, irrigation in segment 2), niwot (day/night steps), russell_2 (litter pool +
nitrogen cycle + anaerobic), a clear-cut and re-planting just before the boundary (ring reset
at death), and a half-hourly year (240 live ring entries, wrapped cursors)."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import _lib
from tests import helpers

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "sipnet_amd", "bin", "sipnet")
REF_BIN = os.path.join(REPO, "oracle", "_ref", "sipnet_ref")
GOLD = os.path.join(helpers.GOLDEN, "restart")
CASES = ["mvp", "niwot", "russell_2", "russell_replant", "halfhourly"]
REL = 2e-11   # our strict fp64 kernel vs glibc: <=1-2 ulp per pow/exp, accumulated over a segment


def case_inputs(case):
    """-> (param path, clim lines, split index)"""
    d = os.path.join(GOLD, case)
    split = int(open(os.path.join(d, "split.txt")).read())
    sm = os.path.join(helpers.GOLDEN, "smoke")
    if case == "mvp":
        return os.path.join(d, "run.param"), open(os.path.join(d, "full.clim")).readlines(), split
    if case == "niwot":
        src, param = os.path.join(sm, "niwot", "sipnet.clim.gz"), os.path.join(sm, "niwot", "sipnet.param")
    elif case == "russell_2":
        src, param = os.path.join(sm, "russell_1", "sipnet.clim.gz"), os.path.join(sm, "russell_2", "sipnet.param")
    elif case == "russell_replant":
        src, param = os.path.join(sm, "russell_1", "sipnet.clim.gz"), os.path.join(sm, "russell_1", "sipnet.param")
    else:
        src = os.path.join(helpers.GOLDEN, "synth", "halfhourly.clim.gz")
        param = os.path.join(REPO, "sipnet_amd", "data", "base_forest.param")
    return param, gzip.open(src, "rt").readlines(), split


def stage(case, w, lines, events_name):
    d = os.path.join(GOLD, case)
    param, _, _ = case_inputs(case)
    shutil.copyfile(param, os.path.join(w, "run.param"))
    shutil.copyfile(os.path.join(d, "sipnet.in"), os.path.join(w, "sipnet.in"))
    open(os.path.join(w, "run.clim"), "w").writelines(lines)
    if events_name is None:
        open(os.path.join(w, "events.in"), "w").write(
            open(os.path.join(d, "events_seg1.in")).read() + open(os.path.join(d, "events_seg2.in")).read())
    else:
        shutil.copyfile(os.path.join(d, events_name), os.path.join(w, "events.in"))


def run(binary, w, *args):
    r = subprocess.run([binary, "-i", "sipnet.in", "-f", "run", *args], cwd=w, capture_output=True,
                       text=True, timeout=600)
    return r


def gold_text(case, name):
    return gzip.open(os.path.join(GOLD, case, name), "rt").read()


def assert_checkpoints_close(mine, ref, rel=REL):
    """two Restart structs: integers exact, doubles to `rel` (scaled by the pool sizes for
    values that are differences of large numbers), ring compared over its live entries"""
    for f in ("processed_steps", "boundary_year", "boundary_day", "trackers_last_year",
              "did_leaf_growth", "did_leaf_fall", "phenology_last_year", "is_alive", "mean_length",
              "mean_start", "mean_last"):
        assert getattr(mine, f) == getattr(ref, f), f
    assert list(mine.flags) == list(ref.flags)
    assert mine.model_version == ref.model_version
    for f in ("boundary_time", "boundary_length", "mean_tot_weight"):
        assert getattr(mine, f) == getattr(ref, f), f
    scale = max(abs(v) for v in ref.envi)
    np.testing.assert_allclose(np.array(mine.envi), np.array(ref.envi), rtol=rel, atol=rel * scale)
    flux_scale = max(abs(v) for v in list(ref.trackers)[:10]) + 1e-30
    for i, name in enumerate(_lib.RT_NAMES):
        a, b = mine.trackers[i], ref.trackers[i]
        tol = rel * max(abs(b), flux_scale if i < 13 or i >= 26 else abs(ref.tracker("totGpp")) + 1e-30)
        assert abs(a - b) <= tol, (name, a, b)
    assert abs(mine.d_till_mod - ref.d_till_mod) <= 1e-15
    assert mine.harvest_frac_removed == pytest.approx(ref.harvest_frac_removed, rel=1e-12, abs=1e-15)
    assert mine.harvest_frac_transferred == pytest.approx(ref.harvest_frac_transferred, rel=1e-12, abs=1e-15)
    live = ref.live_slots()
    w_m = np.array([mine.mean_weights[i] for i in live])
    w_r = np.array([ref.mean_weights[i] for i in live])
    np.testing.assert_array_equal(w_m, w_r)            # the schedule depends on step lengths only
    v_m = np.array([mine.mean_values[i] for i in live])
    v_r = np.array([ref.mean_values[i] for i in live])
    vs = np.abs(v_r).max() + 1e-30
    np.testing.assert_allclose(v_m, v_r, rtol=rel * 50, atol=rel * 50 * vs)
    assert abs(mine.mean_sum - ref.mean_sum) <= rel * 50 * max(vs * 5.0, abs(ref.mean_sum))


@pytest.mark.parametrize("case", CASES)
def test_cli_segments_against_the_reference(case, tmp_path):
    """segment 1 -> checkpoint == the reference's; segment 2 resumed from the REFERENCE's
    checkpoint -> `.out` and events.out text == the reference's, end checkpoint == the reference's;
    resumed from OUR checkpoint -> same text; seg1 + seg2 == continuous (testRestartMVP.c:253-304)"""
    _, lines, k = case_inputs(case)
    w = str(tmp_path)
    # continuous
    stage(case, w, lines, None)
    r = run(CLI, w)
    assert r.returncode == 0, r.stdout + r.stderr
    cont = open(os.path.join(w, "run.out")).read().splitlines()
    # segment 1
    stage(case, w, lines[:k], "events_seg1.in")
    r = run(CLI, w, "--restart-out", "mine1.restart")
    assert r.returncode == 0, r.stdout + r.stderr
    seg1 = open(os.path.join(w, "run.out")).read()
    assert seg1 == gold_text(case, "seg1.out.gz")
    assert open(os.path.join(w, "events.out")).read() == open(os.path.join(GOLD, case, "seg1.events.out")).read()
    mine1 = sa.read_restart(os.path.join(w, "mine1.restart"))
    ref1 = sa.read_restart(os.path.join(GOLD, case, "seg1.restart"))
    assert_checkpoints_close(mine1, ref1)
    # segment 2 from the reference's checkpoint
    stage(case, w, lines[k:], "events_seg2.in")
    shutil.copyfile(os.path.join(GOLD, case, "seg1.restart"), os.path.join(w, "ref1.restart"))
    r = run(CLI, w, "--restart-in", "ref1.restart", "--restart-out", "mine2.restart")
    assert r.returncode == 0, r.stdout + r.stderr
    seg2 = open(os.path.join(w, "run.out")).read()
    assert seg2 == gold_text(case, "seg2.out.gz")
    assert open(os.path.join(w, "events.out")).read() == open(os.path.join(GOLD, case, "seg2.events.out")).read()
    mine2 = sa.read_restart(os.path.join(w, "mine2.restart"))
    ref2 = sa.read_restart(os.path.join(GOLD, case, "seg2.restart"))
    assert mine2.processed_steps == len(lines)      # the count runs on across segments
    assert_checkpoints_close(mine2, ref2)
    # segment 2 from our own checkpoint
    r = run(CLI, w, "--restart-in", "mine1.restart")
    assert r.returncode == 0, r.stdout + r.stderr
    seg2_own = open(os.path.join(w, "run.out")).read()
    assert seg2_own == seg2
    hdr = 1 if "year" in cont[0] else 0
    assert cont == seg1.splitlines() + seg2.splitlines()[hdr:]


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/sipnet_ref not on this box")
@pytest.mark.parametrize("case", ["mvp", "russell_2", "russell_replant", "halfhourly"])
def test_reference_binary_resumes_from_our_checkpoint(case, tmp_path):
    """the drop-in direction: a checkpoint written by this engine is accepted by the reference
    and continues to the reference's own segment-2 output"""
    _, lines, k = case_inputs(case)
    w = str(tmp_path)
    stage(case, w, lines[:k], "events_seg1.in")
    r = run(CLI, w, "--restart-out", "mine1.restart")
    assert r.returncode == 0, r.stdout + r.stderr
    stage(case, w, lines[k:], "events_seg2.in")
    r = run(REF_BIN, w, "--restart-in", "mine1.restart")
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(os.path.join(w, "run.out")).read() == gold_text(case, "seg2.out.gz")


def test_cli_restart_failures_exit_9(tmp_path):
    """testStrictClimateMismatchFails / testModelVersionMismatchFails / flags / missing file"""
    case = "mvp"
    _, lines, k = case_inputs(case)
    w = str(tmp_path)
    ck = os.path.join(GOLD, case, "seg1.restart")
    stage(case, w, lines[k:], "events_seg2.in")
    shutil.copyfile(ck, os.path.join(w, "a.restart"))
    assert run(CLI, w, "--restart-in", "a.restart").returncode == 0
    assert run(CLI, w, "--restart-in", "missing.restart").returncode == 6
    assert run(CLI, w, "--restart-in", "a.restart", "--no-snow").returncode == 9
    open(os.path.join(w, "b.restart"), "w").write(open(ck).read().replace("model_version 2.1.0", "model_version 1.0.0"))
    assert run(CLI, w, "--restart-in", "b.restart").returncode == 9
    # the segment must start after the boundary record
    shutil.copyfile(os.path.join(GOLD, case, "restart_segment2_bad.clim"), os.path.join(w, "run.clim"))
    r = run(CLI, w, "--restart-in", "a.restart")
    assert r.returncode == 9 and "boundary" in r.stdout
    # a late start only warns (testRestartNotNearMidnightWarns)
    shutil.copyfile(os.path.join(GOLD, case, "restart_segment2_late.clim"), os.path.join(w, "run.clim"))
    open(os.path.join(w, "sipnet.in"), "w").write("EVENTS 1\nQUIET 0\n")
    r = run(CLI, w, "--restart-in", "a.restart")
    assert r.returncode == 0 and "time gap" in r.stdout
    # RESTART_IN / RESTART_OUT keys of sipnet.in, and their place in the config dump
    stage(case, w, lines[:k], "events_seg1.in")
    open(os.path.join(w, "sipnet.in"), "w").write("EVENTS 1\nQUIET 1\nRESTART_OUT from_file.restart\nDUMP_CONFIG 1\n")
    assert run(CLI, w).returncode == 0 and os.path.exists(os.path.join(w, "from_file.restart"))
    cfg = open(os.path.join(w, "run.config")).read()
    assert [l for l in cfg.splitlines() if "RESTART_OUT" in l and "INPUT_FILE" in l and "from_file.restart" in l]
    # a checkpoint that ends hours before midnight is written, with a warning
    # (testCheckpointFarFromMidnightWarnsAndWrites)
    shutil.copyfile(os.path.join(GOLD, case, "restart_segment1_not_midnight.clim"), os.path.join(w, "run.clim"))
    open(os.path.join(w, "sipnet.in"), "w").write("EVENTS 1\nQUIET 0\n")
    r = run(CLI, w, "--restart-out", "early.restart")
    assert r.returncode == 0 and "before midnight" in r.stdout and os.path.exists(os.path.join(w, "early.restart"))


def _setup_batch(case, lines, events, members, tmp_path, tag, precision=sa.F64, fast=False):
    p = tmp_path / f"{tag}.clim"
    p.write_text("".join(lines))
    flags = sa.flags_from(**{k: v for k, v in sa.read_config(os.path.join(GOLD, case, "sipnet.in")).items()
                             if k in sa.FLAG_NAMES})
    clim = sa.read_clim(p, gdd=flags[1])
    b = sa.Batch(flags, 1, members.shape[0], precision, fast_math=fast if precision == sa.F64 else None)
    b.set_events(0, events)
    b.set_climate(0, clim)
    b.set_params(0, members)
    return b, clim, flags


def _events(case, flags, base, tmp_path, which):
    d = os.path.join(GOLD, case)
    text = "".join(open(os.path.join(d, f)).read() for f in which)
    p = tmp_path / "ev.in"
    p.write_text(text)
    return sa.read_events(p, flags, base)


@pytest.mark.parametrize("case,precision", [("russell_2", sa.F64), ("halfhourly", sa.F64),
                                            ("halfhourly", sa.F32_MIXED), ("russell_replant", sa.F64)])
def test_batch_export_import_equals_continuous(case, precision, tmp_path, monkeypatch):
    """ensemble: run segment 1, export every member, build a NEW batch for segment 2, import,
    run -> NEE/GPP/ET planes identical to the continuous run (throughput kernel where the
    flags allow it, strict otherwise)"""
    param, lines, k = case_inputs(case)
    flags0 = sa.flags_from(**{kk: v for kk, v in sa.read_config(os.path.join(GOLD, case, "sipnet.in")).items()
                              if kk in sa.FLAG_NAMES})
    base, _ = sa.read_params(param, flags0)
    from sipnet_amd import synth
    M = 96
    members = np.array(list(synth.perturbed_params(base, M, seed=11)))
    members[0] = base
    ev_all = _events(case, flags0, base, tmp_path, ["events_seg1.in", "events_seg2.in"])
    ev1 = _events(case, flags0, base, tmp_path, ["events_seg1.in"])
    ev2 = _events(case, flags0, base, tmp_path, ["events_seg2.in"])
    T = len(lines)

    b, clim, flags = _setup_batch(case, lines, ev_all, members, tmp_path, "full", precision, fast=True)
    b.setup()
    cont = b.run(0, T)[0].cpu().numpy()
    ok = b.get_status() == 0
    b.close()

    b1, _, _ = _setup_batch(case, lines[:k], ev1, members, tmp_path, "s1", precision, fast=True)
    b1.setup()
    s1 = b1.run(0, k)[0].cpu().numpy()
    cks = [b1.export_restart(0, m, k) for m in range(M)]
    b1.close()
    ref1 = sa.read_restart(os.path.join(GOLD, case, "seg1.restart"))
    assert (cks[0].mean_start, cks[0].mean_last) == (ref1.mean_start, ref1.mean_last)
    if precision == sa.F64:
        # member 0 is the reference's own parameter set: the throughput kernel's checkpoint
        # is the reference's to fast-math accuracy
        np.testing.assert_allclose(np.array(cks[0].envi), np.array(ref1.envi), rtol=1e-9,
                                   atol=1e-9 * max(abs(v) for v in ref1.envi))

    b2, clim2, _ = _setup_batch(case, lines[k:], ev2, members, tmp_path, "s2", precision, fast=True)
    assert sa.check_restart(cks[0], flags, clim2) & _lib.RESTART_WARN_TIME_GAP == 0
    b2.set_resume(0, cks[0])
    b2.setup()
    b2.import_restart(0, cks)
    s2 = b2.run(0, T - k)[0].cpu().numpy()
    b2.close()
    for name, c, a, z in zip(("nee", "gpp", "et"), cont, s1, s2):
        got = np.concatenate([a, z], axis=0)[:, ok]
        want = c[:, ok]
        if case == "russell_replant":
            # re-planted members resume from a reset ring whose partial eviction weights
            # differ from the continuous run's by rounding (ring epochs)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
        else:
            np.testing.assert_array_equal(got, want, err_msg=name)


def test_batch_import_relays_a_reset_ring_onto_the_site_layout(tmp_path):
    """one site, two histories: member 0 was clear-cut and re-planted (ring reset at death), the
    others never died.  Importing the re-planted member onto the never-died layout keeps its
    trajectory (its older entries are zero in any layout); a ring that cannot be expressed is
    refused with code 9."""
    case = "russell_replant"
    param, lines, k = case_inputs(case)
    flags = sa.flags_from()
    base, _ = sa.read_params(param, flags)
    members = np.stack([base, base])
    ev1 = _events(case, flags, base, tmp_path, ["events_seg1.in"])
    ev2 = _events(case, flags, base, tmp_path, ["events_seg2.in"])
    T = len(lines)
    # segment 1 twice: with the clear-cut (member dies, is re-planted) and without events
    b, _, _ = _setup_batch(case, lines[:k], ev1, members, tmp_path, "a")
    b.setup()
    b.run(0, k)
    replanted = b.export_restart(0, 0, k)
    b.close()
    till_only = [e for e in ev1 if e.type == 4]
    b, _, _ = _setup_batch(case, lines[:k], till_only, members, tmp_path, "b")
    b.setup()
    b.run(0, k)
    intact = b.export_restart(0, 1, k)
    b.close()
    assert len(replanted.live_slots()) == 25 and len(intact.live_slots()) in (40, 41)
    assert replanted.d_till_mod == intact.d_till_mod
    # reference trajectory of the re-planted member: resumed alone on its own layout
    b, _, _ = _setup_batch(case, lines[k:], ev2, members[:1], tmp_path, "c")
    b.set_resume(0, replanted)
    b.setup()
    b.import_restart(0, [replanted])
    alone = b.run(0, T - k)[0].cpu().numpy()[:, :, 0]
    b.close()
    # both members in one batch on the never-died layout
    b, _, _ = _setup_batch(case, lines[k:], ev2, members, tmp_path, "d")
    b.set_resume(0, intact)
    b.setup()
    b.import_restart(0, [replanted, intact])
    both = b.run(0, T - k)[0].cpu().numpy()
    for a, c in zip(alone, both):
        np.testing.assert_allclose(c[:, 0], a, rtol=1e-10, atol=1e-13)
    # the other way round cannot work: the intact ring has non-zero entries older than the
    # re-planted layout can hold
    b.set_resume(0, replanted)
    b.setup()
    with pytest.raises(sa.SipnetError) as e:
        b.import_restart(0, [replanted, intact])
    assert e.value.code == _lib.ERR_RESTART and "layout" in str(e.value)
    # members of a site share one forcing history
    b.set_resume(0, intact)
    b.setup()
    other = sa.read_restart(os.path.join(GOLD, "mvp", "seg1.restart"))
    with pytest.raises(sa.SipnetError) as e:
        b.import_restart(0, [intact, other])
    assert e.value.code == _lib.ERR_RESTART
    # survival.isAlive must agree with the pools
    intact.is_alive = 0
    with pytest.raises(sa.SipnetError) as e:
        b.import_restart(0, [intact])
    assert e.value.code == _lib.ERR_RESTART and "isAlive" in str(e.value)
    # export needs the state to be where the caller says it is
    with pytest.raises(sa.SipnetError):
        b.export_restart(0, 0, 5)
    b.close()



SITE_CASES = ["niwot", "russell_2", "russell_replant", "halfhourly"]


def _stage_sites(root, seg, extra_in):
    """one run directory per case under root/<case>: segment 1 or 2 of its forcing; sipnet.in += extra_in(case)"""
    for case in SITE_CASES:
        _, lines, k = case_inputs(case)
        w = os.path.join(root, case)
        os.makedirs(w, exist_ok=True)
        stage(case, w, lines[:k] if seg == 1 else lines[k:], f"events_seg{seg}.in")
        with open(os.path.join(w, "sipnet.in"), "a") as f:
            f.write(extra_in(case))
    open(os.path.join(root, "runs.txt"), "w").write("\n".join(SITE_CASES) + "\n")


def run_sites(root, *args):
    return subprocess.run([CLI, "--sites", "runs.txt", "-i", "sipnet.in", "-f", "run", *args], cwd=root, capture_output=True,
                          text=True, timeout=900)


def test_sites_with_restart_checkpoints_two_invocations(tmp_path):
    """SDA at the process boundary: `sipnet --sites` over one directory per member, RESTART_OUT / RESTART_IN in each
    directory's sipnet.in (frontend.c:164-209, sipnet.c:1963-1989).  Invocation 1 (segment-1 directories of four
    cases: three flag sets, three step lengths, in shared batches) writes the files and checkpoints of
    test_cli_segments_against_the_reference; invocation 2 resumes every directory from the REFERENCE's checkpoint and
    gives the reference's segment-2 text and end checkpoints; resumed from OUR checkpoints it gives the same text"""
    root = str(tmp_path)
    _stage_sites(root, 1, lambda case: "RESTART_OUT = mine1.restart\n")
    r = run_sites(root)
    assert r.returncode == 0, r.stdout + r.stderr
    for case in SITE_CASES:
        w = os.path.join(root, case)
        assert open(os.path.join(w, "run.out")).read() == gold_text(case, "seg1.out.gz"), case
        assert open(os.path.join(w, "events.out")).read() == open(os.path.join(GOLD, case, "seg1.events.out")).read(), case
        assert_checkpoints_close(sa.read_restart(os.path.join(w, "mine1.restart")), sa.read_restart(os.path.join(GOLD, case, "seg1.restart")))
        shutil.copyfile(os.path.join(w, "mine1.restart"), os.path.join(root, f"{case}.mine1"))
    # segment 2 from the reference's checkpoints (RESTART_IN relative to the run directory, RESTART_OUT absolute)
    _stage_sites(root, 2, lambda case: f"RESTART_IN = ref1.restart\nRESTART_OUT = {root}/{case}.mine2\n")
    for case in SITE_CASES:
        shutil.copyfile(os.path.join(GOLD, case, "seg1.restart"), os.path.join(root, case, "ref1.restart"))
    r = run_sites(root)
    assert r.returncode == 0, r.stdout + r.stderr
    seg2 = {}
    for case in SITE_CASES:
        w = os.path.join(root, case)
        seg2[case] = open(os.path.join(w, "run.out")).read()
        assert seg2[case] == gold_text(case, "seg2.out.gz"), case
        assert open(os.path.join(w, "events.out")).read() == open(os.path.join(GOLD, case, "seg2.events.out")).read(), case
        mine2 = sa.read_restart(os.path.join(root, f"{case}.mine2"))
        assert mine2.processed_steps == len(case_inputs(case)[1])
        assert_checkpoints_close(mine2, sa.read_restart(os.path.join(GOLD, case, "seg2.restart")))
    # ... and from our own
    _stage_sites(root, 2, lambda case: "RESTART_IN = own1.restart\n")
    for case in SITE_CASES:
        shutil.copyfile(os.path.join(root, f"{case}.mine1"), os.path.join(root, case, "own1.restart"))
    r = run_sites(root)
    assert r.returncode == 0, r.stdout + r.stderr
    for case in SITE_CASES:
        assert open(os.path.join(root, case, "run.out")).read() == seg2[case], case
    # the drop-in direction: the REFERENCE binary resumes from a checkpoint `--sites` wrote (each directory's sipnet.in
    # still names own1.restart = the copy of mine1.restart)
    if os.path.exists(REF_BIN):
        for case in ("russell_2", "russell_replant", "halfhourly"):
            w = os.path.join(root, case)
            os.remove(os.path.join(w, "run.out"))
            r = run(REF_BIN, w)
            assert r.returncode == 0, r.stdout + r.stderr
            assert open(os.path.join(w, "run.out")).read() == gold_text(case, "seg2.out.gz"), case


def test_sites_members_resume_from_their_own_checkpoints_in_one_site(tmp_path):
    """three members of ONE site (same forcing, other parameters), two cycles: `--sites` twice with checkpoints
    == `--ensemble-params` in one piece.  A fourth directory whose checkpoint comes from another history (other year
    counters / GDD) must not share their site -- and still runs"""
    root = str(tmp_path)
    case = "halfhourly"
    _, lines, k = case_inputs(case)
    amax = (7.2, 8.4, 9.1)
    def stage_members(seg, extra):
        for m, a in enumerate(amax):
            w = os.path.join(root, f"m{m}")
            os.makedirs(w, exist_ok=True)
            stage(case, w, lines[:k] if seg == 1 else lines[k:], f"events_seg{seg}.in")
            txt = [(f"aMax {a}" if l.split() and l.split()[0] == "aMax" else l) for l in open(os.path.join(w, "run.param")).read().splitlines()]
            open(os.path.join(w, "run.param"), "w").write("\n".join(txt) + "\n")
            with open(os.path.join(w, "sipnet.in"), "a") as f:
                f.write(extra)
    stage_members(1, "RESTART_OUT = cycle1.restart\n")
    open(os.path.join(root, "runs.txt"), "w").write("m0\nm1\nm2\n")
    r = run_sites(root)
    assert r.returncode == 0 and "1 site(s) x up to 3 member(s)" in r.stdout, r.stdout + r.stderr
    seg1 = [open(os.path.join(root, f"m{m}", "run.out")).read() for m in range(3)]
    stage_members(2, "RESTART_IN = cycle1.restart\n")
    # an intruder: same segment-2 forcing, but a checkpoint of a DIFFERENT history (niwot's would have other flags; use
    # member 0's with the year-to-date GDD moved) -- it must get a site of its own
    w = os.path.join(root, "odd")
    shutil.copytree(os.path.join(root, "m0"), w)
    ck = sa.read_restart(os.path.join(w, "cycle1.restart"))
    ck.trackers[_lib.RT_NAMES.index("gdd")] += 1.0
    sa.write_restart(os.path.join(w, "cycle1.restart"), ck)
    open(os.path.join(root, "runs.txt"), "w").write("m0\nm1\nodd\nm2\n")
    r = run_sites(root)
    assert r.returncode == 0 and "2 site(s) x up to 3 member(s)" in r.stdout, r.stdout + r.stderr
    seg2 = [open(os.path.join(root, f"m{m}", "run.out")).read() for m in range(3)]
    # the same three members in one piece
    w = os.path.join(root, "whole")
    os.makedirs(w)
    stage(case, w, lines, None)
    open(os.path.join(w, "members.txt"), "w").write("aMax\n" + "\n".join(str(a) for a in amax) + "\n")
    r = subprocess.run([CLI, "-i", "sipnet.in", "-f", "run", "--ensemble-params", "members.txt", "--math", "strict"], cwd=w,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for m in range(3):
        whole = open(os.path.join(w, f"run.{m}.out")).read().splitlines()
        hdr = 1 if "year" in whole[0] else 0
        assert whole == seg1[m].splitlines() + seg2[m].splitlines()[hdr:], m
    assert seg2[0] != seg2[2]
