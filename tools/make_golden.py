#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (build container only).

Needs /root/reference and oracle/_ref (make -C oracle ref).  What it commits is
data only: the input/expected-output files of the reference's own smoke tests
(tests/smoke/*), and full-precision per-step records obtained by running the
reference's step loop in-process through oracle/ref_harness.c.  No reference
source text is stored.

    python tools/make_golden.py
"""
import ctypes as C
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
REF_SO = os.path.join(REPO, "oracle", "_ref", "libsipnet_ref.so")
NREC = 36


def gz_copy(src, dst):
    with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
        shutil.copyfileobj(fi, fo)


def copy_smoke():
    for case in ["niwot", "russell_1", "russell_2", "russell_3"]:
        s = os.path.join(REF, "tests", "smoke", case)
        d = os.path.join(GOLD, "smoke", case)
        os.makedirs(d, exist_ok=True)
        for f in ["sipnet.in", "sipnet.param", "events.in", "events.out", "sipnet.config"]:
            shutil.copyfile(os.path.join(s, f), os.path.join(d, f))
        gz_copy(os.path.join(s, "sipnet.out"), os.path.join(d, "sipnet.out.gz"))
        if case in ("niwot", "russell_1"):  # russell_2/3 use russell_1's forcing
            gz_copy(os.path.join(s, "sipnet.clim"), os.path.join(d, "sipnet.clim.gz"))
    a = open(os.path.join(REF, "tests/smoke/russell_1/sipnet.clim"), "rb").read()
    for c in ("russell_2", "russell_3", "russell_4"):
        assert open(os.path.join(REF, f"tests/smoke/{c}/sipnet.clim"), "rb").read() == a
    # russell_4 (events off, GDD off, soil-temperature phenology, snow flag off) is skipped by the
    # reference's own smoke driver (`skip` marker; its committed outputs predate the current
    # model).  Inputs are copied; the expected outputs come from the reference binary built here.
    import tempfile
    s = os.path.join(REF, "tests", "smoke", "russell_4")
    d = os.path.join(GOLD, "smoke", "russell_4")
    os.makedirs(d, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mkgold_r4_")
    for f in ["sipnet.in", "sipnet.param", "events.in", "sipnet.clim"]:
        shutil.copyfile(os.path.join(s, f), os.path.join(tmp, f))
    run_ref_cli(tmp, ["-i", "sipnet.in"])
    for f in ["sipnet.in", "sipnet.param", "events.in"]:
        shutil.copyfile(os.path.join(s, f), os.path.join(d, f))
    shutil.copyfile(os.path.join(tmp, "sipnet.config"), os.path.join(d, "sipnet.config"))
    gz_copy(os.path.join(tmp, "sipnet.out"), os.path.join(d, "sipnet.out.gz"))
    assert not os.path.exists(os.path.join(tmp, "events.out"))   # EVENTS = 0: none is written
    open(os.path.join(d, "events.out"), "wb").close()
    shutil.rmtree(tmp)


def ref_run(flags, param_file, clim_file, events_file, raw_members=None):
    """Run the real reference in a child process (it keeps process-global state).
    -> rec[n_members][n_steps][36]"""
    code = r"""
import ctypes as C, numpy as np, sys, pickle
flags, param_file, clim_file, events_file, raw_path, out_path, so = pickle.load(open(sys.argv[1],'rb'))
ref = C.CDLL(so)
fl = (C.c_int*12)(*flags)
n = ref.ref_init(fl, param_file.encode(), clim_file.encode(), events_file.encode(), b"/dev/null")
NP = ref.ref_num_params(); NR = ref.ref_rec_len()
base = np.zeros(NP); ref.ref_get_base_params(base.ctypes.data_as(C.c_void_p))
raw = np.load(raw_path) if raw_path else base[None,:]
out = np.zeros((raw.shape[0], n, NR))
for m in range(raw.shape[0]):
    r = np.ascontiguousarray(raw[m])
    ref.ref_run_member(r.ctypes.data_as(C.c_void_p), out[m].ctypes.data_as(C.c_void_p), None, None, None)
np.save(out_path, out)
"""
    import pickle, tempfile
    tmp = tempfile.mkdtemp(prefix="mkgold_")
    raw_path = None
    if raw_members is not None:
        raw_path = os.path.join(tmp, "raw.npy")
        np.save(raw_path, raw_members)
    out_path = os.path.join(tmp, "out.npy")
    job = os.path.join(tmp, "job.pkl")
    pickle.dump((list(flags), param_file, clim_file, events_file, raw_path, out_path, REF_SO),
                open(job, "wb"))
    subprocess.check_call([sys.executable, "-c", code, job])
    out = np.load(out_path)
    shutil.rmtree(tmp)
    return out


def ref_warning_counts(flags, param_file, clim_file, raw_members, first_steps=None):
    """The reference's OWN per-step self-check warnings, counted per member: the real reference (not
    quiet) runs every member in a child process whose stdout is parsed for
      "[WARNING] ... Non-negative stock constraint applied for ..."   ensureNonNegative(), sipnet.c:1346-1356
      "[WARNING] ... Carbon / Nitrogen balance check failed ..."      checkBalance(), balance.c:149-163
    -> (n_clamp_warn[n_members], n_balance_warn[n_members])"""
    code = r"""
import ctypes as C, numpy as np, sys, pickle, os
flags, param_file, clim_file, raw_path, so = pickle.load(open(sys.argv[1],'rb'))
ref = C.CDLL(so); libc = C.CDLL(None)
fl = (C.c_int*12)(*flags)
n = ref.ref_init(fl, param_file.encode(), clim_file.encode(), b"/nonexistent/events.in", b"/dev/null")
ref.ref_set_quiet(0)
raw = np.load(raw_path)
for m in range(raw.shape[0]):
    libc.fflush(None); os.write(1, b"\n@@MEMBER %d\n" % m)
    r = np.ascontiguousarray(raw[m])
    ref.ref_run_member(r.ctypes.data_as(C.c_void_p), None, None, None, None)
libc.fflush(None)
"""
    import pickle, tempfile
    tmp = tempfile.mkdtemp(prefix="mkgold_")
    raw_path = os.path.join(tmp, "raw.npy")
    np.save(raw_path, raw_members)
    job = os.path.join(tmp, "job.pkl")
    pickle.dump((list(flags), param_file, clim_file, raw_path, REF_SO), open(job, "wb"))
    out = subprocess.run([sys.executable, "-c", code, job], check=True, capture_output=True).stdout.decode(errors="replace")
    shutil.rmtree(tmp)
    clamp, bal = np.zeros(len(raw_members), dtype=np.int64), np.zeros(len(raw_members), dtype=np.int64)
    m = -1
    for line in out.split("\n"):
        if line.startswith("@@MEMBER"):
            m = int(line.split()[1])
        elif "[WARNING]" in line and m >= 0:
            if "Non-negative stock constraint applied" in line:
                clamp[m] += 1
            elif "balance check failed" in line:
                bal[m] += 1
    return clamp, bal


def warning_counts():
    """tests/golden/synth/ref_warning_counts.json: the reference's warning lines on the 16 special
    members (synthetic half-hourly year) and on a 1e11-gC stand (20 days), for
    tests/test_oracle_golden.py to hold the oracle's sipo_diag counters against"""
    import json
    import sipnet_amd as sa
    from sipnet_amd import synth
    from sipnet_amd.config import param_index as pi
    d = os.path.join(GOLD, "synth")
    flags = sa.flags_from()
    base_file = os.path.join(REPO, "sipnet_amd", "data", "base_forest.param")
    tmp = tempfile.mkdtemp(prefix="mkgold_")
    clim_path = os.path.join(tmp, "hh.clim")
    with gzip.open(os.path.join(d, "halfhourly.clim.gz"), "rb") as fi, open(clim_path, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    members = np.load(os.path.join(d, "members_raw.npy"))
    c16, b16 = ref_warning_counts(flags, base_file, clim_path, members)
    # the heavy stand of tests/test_gpu_full.py: one ulp of its carbon total exceeds the balance threshold
    base, _ = sa.read_params(base_file, flags)
    heavy = synth.perturbed_params(base, 8)
    heavy[:, pi("plantWoodInit")] = 1e11
    raw = synth.half_hourly_year_raw(150 * 48 + 48 * 20)
    raw = {k: v[150 * 48:] for k, v in raw.items()}
    clim20 = os.path.join(tmp, "d20.clim")
    synth.write_clim(clim20, synth.round_like_file(raw))
    ch, bh = ref_warning_counts(flags, base_file, clim20, heavy)
    shutil.rmtree(tmp)
    out = {"what": "lines the real reference (oracle/_ref/libsipnet_ref.so, not quiet) printed per member: "
                   "ensureNonNegative() warnings (sipnet.c:1346-1356) and checkBalance() failures (balance.c:149-163)",
           "special16": {"n_clamp_warn": c16.tolist(), "n_balance_warn": b16.tolist()},
           "wood1e11_20days_from_day150": {"n_clamp_warn": ch.tolist(), "n_balance_warn": bh.tolist()}}
    json.dump(out, open(os.path.join(d, "ref_warning_counts.json"), "w"), indent=1)
    print("warning counts:", out["special16"], out["wood1e11_20days_from_day150"])


def decimate_index(n, head=300, tail=300, every=7):
    idx = set(range(min(head, n))) | set(range(max(0, n - tail), n)) | set(range(0, n, every))
    return np.array(sorted(idx), dtype=np.int32)


def smoke_records():
    import sipnet_amd as sa
    out = {}
    for case in ["niwot", "russell_1", "russell_2", "russell_3", "russell_4"]:
        s = os.path.join(REF, "tests", "smoke", case)
        cfg = sa.read_config(os.path.join(s, "sipnet.in"))
        flags = [cfg[n] for n in sa.FLAG_NAMES]
        rec = ref_run(flags, os.path.join(s, "sipnet.param"), os.path.join(s, "sipnet.clim"),
                      os.path.join(s, "events.in"))[0]
        idx = decimate_index(rec.shape[0])
        out[f"{case}_idx"] = idx
        out[f"{case}_rec"] = rec[idx]
        out[f"{case}_final"] = rec[-1]
        print(case, rec.shape, "kept", len(idx))
    np.savez_compressed(os.path.join(GOLD, "ref_smoke_records.npz"), **out)


def special_members(base):
    """A small ensemble that walks the rare branches: phenology, mortality, drought."""
    import sipnet_amd as sa
    from sipnet_amd.config import param_index as pi
    from sipnet_amd import synth
    mem = list(synth.perturbed_params(base, 6, seed=synth.SEED_PARAMS))
    def variant(**kw):
        p = base.copy()
        for k, v in kw.items():
            assert pi(k) >= 0, k
            p[pi(k)] = v
        mem.append(p)
    variant(leafGrowth=120.0, fracLeafFall=0.6, gddLeafOn=300.0, leafOffDay=280.0)   # deciduous
    variant(leafGrowth=60.0, fracLeafFall=1.0, gddLeafOn=150.0, leafOffDay=250.0, laiInit=0.0)
    variant(baseVegResp=2.0, plantWoodInit=300.0)            # respires itself to death (step 10074)
    variant(plantWoodInit=0.0)                                # dead from the start
    variant(soilWHC=0.5, soilWFracInit=0.1)                   # drought / tiny bucket
    variant(snowInit=8.0, frozenSoilEff=0.5, frozenSoilFolREff=0.5, frozenSoilThreshold=1.0)
    variant(soilRespMoistEffect=1.7, dVpdExp=1.5)             # non-trivial exponents
    variant(fineRootFrac=0.0005, fineRootAllocation=0.0, fineRootTurnoverRate=0.9)  # root deficit rerouting
    variant(laiInit=0.05, leafTurnoverRate=0.9, leafAllocation=0.01)                 # leaf deficit rerouting
    variant(leafGrowth=5000.0, fracLeafFall=0.3, gddLeafOn=200.0, leafOffDay=270.0)  # leaf-on C-limited
    return np.array(mem)


def synthetic():
    import sipnet_amd as sa
    from sipnet_amd import synth
    d = os.path.join(GOLD, "synth")
    os.makedirs(d, exist_ok=True)
    raw = synth.round_like_file(synth.half_hourly_year_raw())
    clim_path = os.path.join(d, "halfhourly.clim")
    synth.write_clim(clim_path, raw)
    flags = sa.flags_from()
    base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
    members = special_members(base)
    np.save(os.path.join(d, "members_raw.npy"), members)
    # the reference reads the files; the base param file only seeds `params`, every
    # member vector is laid over it by the harness
    rec = ref_run(flags, os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), clim_path,
                  "/nonexistent/events.in", members)
    n = rec.shape[1]
    idx = decimate_index(n, head=600, tail=300, every=5)
    np.savez_compressed(os.path.join(d, "ref_synth.npz"), idx=idx,
                        nee=rec[:, idx, 0], gpp=rec[:, idx, 1], et=rec[:, idx, 2],
                        final=rec[:, -1, :], sum_nee=rec[:, :, 0].sum(1),
                        leaf=rec[:, idx, 15], snow=rec[:, idx, 19], wood=rec[:, idx, 14])
    gz_copy(clim_path, clim_path + ".gz")
    os.remove(clim_path)
    print("synthetic members", members.shape, "steps", n, "kept", len(idx))
    print("  died/zero-wood finals:", rec[:, -1, 14])
    print("  final snow:", rec[:, -1, 19])

    # variable-flag synthetic: a 3-hourly two-month window with every optional flag on
    # (the russell_2 parameter file carries the N-cycle parameters)
    flags2 = sa.flags_from(litterPool=1, nitrogenCycle=1, anaerobic=1, growthResp=1, leafWater=1,
                           flooding=1, carbonSaturation=1)
    s = os.path.join(REF, "tests", "smoke", "russell_2")
    p2, seen = sa.read_params(os.path.join(s, "sipnet.param"), sa.flags_from(litterPool=1, nitrogenCycle=1, anaerobic=1))
    # the extra flags need a few more parameters than russell_2's file carries
    from sipnet_amd.config import param_index as pi
    extra = dict(growthRespFrac=0.2, leafPoolDepth=0.1, waterDrainFrac=0.3, soilCSaturation=6000.0)
    lines = open(os.path.join(s, "sipnet.param")).read()
    allp = os.path.join(d, "allflags.param")
    with open(allp, "w") as fh:
        fh.write(lines)
        for k, v in extra.items():
            if not seen[pi(k)]:
                fh.write(f"{k} {v}\n")
    rec2 = ref_run(flags2, allp, os.path.join(s, "sipnet.clim"), os.path.join(s, "events.in"))[0]
    idx2 = decimate_index(rec2.shape[0])
    np.savez_compressed(os.path.join(d, "ref_allflags.npz"), idx=idx2, rec=rec2[idx2],
                        flags=np.array(flags2))
    print("allflags", rec2.shape)


REF_BIN = os.path.join(REPO, "oracle", "_ref", "sipnet_ref")


def run_ref_cli(workdir, args):
    r = subprocess.run([REF_BIN] + args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    assert r.returncode == 0, (args, r.stdout[-2000:])
    return r.stdout


def day_split_index(lines, ycol, approx):
    """first row index >= approx that starts a new day"""
    key = lambda ln: tuple(ln.split()[ycol:ycol + 2])
    k = approx
    while key(lines[k]) == key(lines[k - 1]):
        k += 1
    return k


def restart_case(name, param_src, clim_lines, split, in_text, events_full, events_seg1, events_seg2,
                 keep_inputs):
    """Run the real reference continuous / segment 1 / segment 2 (resuming from its own
    checkpoint) and keep: seg1.restart, seg2.restart, seg1.out, seg2.out (+ events.out)."""
    import tempfile
    d = os.path.join(GOLD, "restart", name)
    os.makedirs(d, exist_ok=True)
    with tempfile.TemporaryDirectory() as w:
        shutil.copyfile(param_src, os.path.join(w, "run.param"))
        open(os.path.join(w, "sipnet.in"), "w").write(in_text)
        def go(lines, events, extra):
            open(os.path.join(w, "run.clim"), "w").writelines(lines)
            open(os.path.join(w, "events.in"), "w").write(events)
            run_ref_cli(w, ["-i", "sipnet.in", "-f", "run"] + extra)
        go(clim_lines, events_full, [])
        shutil.copyfile(os.path.join(w, "run.out"), os.path.join(w, "continuous.out"))
        shutil.copyfile(os.path.join(w, "events.out"), os.path.join(w, "continuous.events.out"))
        go(clim_lines[:split], events_seg1, ["--restart-out", "seg1.restart"])
        shutil.copyfile(os.path.join(w, "run.out"), os.path.join(w, "seg1.out"))
        shutil.copyfile(os.path.join(w, "events.out"), os.path.join(w, "seg1.events.out"))
        go(clim_lines[split:], events_seg2, ["--restart-in", "seg1.restart", "--restart-out", "seg2.restart"])
        shutil.copyfile(os.path.join(w, "run.out"), os.path.join(w, "seg2.out"))
        shutil.copyfile(os.path.join(w, "events.out"), os.path.join(w, "seg2.events.out"))
        # the property the reference's own test asserts (testRestartMVP.c:253-304)
        cont = open(os.path.join(w, "continuous.out")).read().splitlines()
        s1 = open(os.path.join(w, "seg1.out")).read().splitlines()
        s2 = open(os.path.join(w, "seg2.out")).read().splitlines()
        hdr = 1 if "year" in cont[0] else 0
        assert cont == s1 + s2[hdr:], f"{name}: reference segmented run differs from continuous"
        for f in ["seg1.restart", "seg2.restart", "seg1.events.out", "seg2.events.out"]:
            shutil.copyfile(os.path.join(w, f), os.path.join(d, f))
        gz_copy(os.path.join(w, "seg1.out"), os.path.join(d, "seg1.out.gz"))
        gz_copy(os.path.join(w, "seg2.out"), os.path.join(d, "seg2.out.gz"))
    open(os.path.join(d, "sipnet.in"), "w").write(in_text)
    open(os.path.join(d, "events_seg1.in"), "w").write(events_seg1)
    open(os.path.join(d, "events_seg2.in"), "w").write(events_seg2)
    open(os.path.join(d, "split.txt"), "w").write(f"{split}\n")
    if keep_inputs:
        shutil.copyfile(param_src, os.path.join(d, "run.param"))
        open(os.path.join(d, "full.clim"), "w").writelines(clim_lines)
    print("restart case", name, "split", split, "of", len(clim_lines))


def split_events(text, clim_lines, split, ycol):
    """events before / from the first record of segment 2"""
    y, dd = (int(v) for v in clim_lines[split].split()[ycol:ycol + 2])
    a, b = [], []
    for ln in text.splitlines(True):
        t = ln.split()
        if len(t) < 3:
            continue
        (a if (int(t[0]), int(t[1])) < (y, dd) else b).append(ln)
    return "".join(a), "".join(b)


def restart_cases():
    R = os.path.join(REF, "tests", "sipnet", "test_restart_infrastructure")
    rd = lambda f: open(os.path.join(R, f)).read()
    # 1. the reference's own restart test case (testRestartMVP.c): its data files
    d = os.path.join(GOLD, "restart", "mvp")
    os.makedirs(d, exist_ok=True)
    for f in ["restart_segment2_bad.clim", "restart_segment2_late.clim",
              "restart_segment1_not_midnight.clim"]:
        shutil.copyfile(os.path.join(R, f), os.path.join(d, f))
    full = open(os.path.join(R, "restart_full.clim")).readlines()
    n1 = len(open(os.path.join(R, "restart_segment1.clim")).readlines())
    assert full[:n1] == open(os.path.join(R, "restart_segment1.clim")).readlines()
    assert full[n1:] == open(os.path.join(R, "restart_segment2.clim")).readlines()
    restart_case("mvp", os.path.join(R, "restart.param"), full, n1, "EVENTS 1\nQUIET 1\n",
                 rd("events_base.in"), rd("events_segment1.in"), rd("events_segment2.in"), True)

    sm = os.path.join(REF, "tests", "smoke")
    # 2. niwot: default flags, day/night steps, no events
    lines = open(os.path.join(sm, "niwot", "sipnet.clim")).readlines()
    k = day_split_index(lines, 1, 2700)
    restart_case("niwot", os.path.join(sm, "niwot", "sipnet.param"), lines, k,
                 "PRINT_HEADER = 0\nQUIET = 1\n", "", "", "", False)
    # 3. russell_2: litter pool + nitrogen cycle + anaerobic, irrigation and fertiliser events
    lines = open(os.path.join(sm, "russell_1", "sipnet.clim")).readlines()
    k = day_split_index(lines, 0, 1500)
    ev = open(os.path.join(sm, "russell_2", "events.in")).read()
    e1, e2 = split_events(ev, lines, k, 0)
    restart_case("russell_2", os.path.join(sm, "russell_2", "sipnet.param"), lines, k,
                 "EVENTS = 1\nLITTER_POOL = 1\nNITROGEN_CYCLE = 1\nANAEROBIC = 1\nQUIET = 1\n",
                 ev, e1, e2, False)
    # 4. clear-cut and re-planting shortly before the boundary: the member dies (ring reset,
    #    sipnet.c:1757), is re-planted, and the checkpoint is taken while the ring still
    #    holds its reset entry; tillage modifier still decaying across the boundary
    ev = ("2016 150 harv 1.0 1.0 0.0 0.0\n2016 158 till 0.4\n"
          "2016 160 plant 20 300 40 60\n2016 170 irrig 2.5 0\n")
    k = day_split_index(lines, 0, 8 * 161 + 4)
    e1, e2 = split_events(ev, lines, k, 0)
    restart_case("russell_replant", os.path.join(sm, "russell_1", "sipnet.param"), lines, k,
                 "EVENTS = 1\nQUIET = 1\n", ev, e1, e2, False)
    # 5. half-hourly synthetic year: 240 live ring entries, cursors wrapped several times
    import gzip as gz
    lines = gz.open(os.path.join(GOLD, "synth", "halfhourly.clim.gz"), "rt").readlines()
    k = day_split_index(lines, 0, 48 * 200 + 7)
    restart_case("halfhourly", os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), lines, k,
                 "PRINT_HEADER = 0\nQUIET = 1\n", "", "", "", False)


def events_infra_case():
    """data files of the reference's events.out format test and of its event-file parser tests
    (tests/sipnet/test_events_infrastructure, tests/sipnet/test_bugfixes)"""
    d = os.path.join(GOLD, "events_infra")
    os.makedirs(d, exist_ok=True)
    R = os.path.join(REF, "tests", "sipnet", "test_events_infrastructure")
    for f in os.listdir(R):
        if f.endswith(".in") or f.endswith(".out"):
            shutil.copyfile(os.path.join(R, f), os.path.join(d, f))
    R = os.path.join(REF, "tests", "sipnet", "test_bugfixes")
    for f in os.listdir(R):
        if f.endswith(".in"):
            shutil.copyfile(os.path.join(R, f), os.path.join(d, f))


def sipnet_infra_case():
    """data files of the reference's input-reader tests (tests/sipnet/test_sipnet_infrastructure)"""
    d = os.path.join(GOLD, "sipnet_infra")
    os.makedirs(d, exist_ok=True)
    R = os.path.join(REF, "tests", "sipnet", "test_sipnet_infrastructure")
    for f in os.listdir(R):
        if f.endswith((".clim", ".param", ".exp")):
            shutil.copyfile(os.path.join(R, f), os.path.join(d, f))


def balance_case():
    """data files of the reference's mass-balance test (tests/sipnet/test_modeling/testBalance.c)"""
    R = os.path.join(REF, "tests", "sipnet", "test_modeling")
    d = os.path.join(GOLD, "balance")
    os.makedirs(d, exist_ok=True)
    for f in ["balance.clim", "balance.param", "events_leaf.in"]:
        shutil.copyfile(os.path.join(R, f), os.path.join(d, f))


def debug_log_cases():
    """the three --debug-log files (debug_log.c) of the reference binary on two smoke cases:
    header + a decimated set of rows (row numbers in <case>_rows.txt), gzipped"""
    import tempfile
    d = os.path.join(GOLD, "debug_log")
    os.makedirs(d, exist_ok=True)
    for case in ["niwot", "russell_2"]:
        src = os.path.join(REF, "tests", "smoke", case)
        tmp = tempfile.mkdtemp(prefix="mkgold_dbg_")
        for f in ["sipnet.in", "sipnet.param", "sipnet.clim", "events.in"]:
            shutil.copyfile(os.path.join(src, f), os.path.join(tmp, f))
        run_ref_cli(tmp, ["-i", "sipnet.in", "--debug-log", "dbg"])
        first = open(os.path.join(tmp, "dbg_envi.log")).readline()
        hdr = 1 if first.startswith("year") else 0  # headers only with PRINT_HEADER (sipnet.c:1959)
        n = sum(1 for _ in open(os.path.join(tmp, "dbg_envi.log"))) - hdr
        idx = decimate_index(n, head=40, tail=40, every=53)
        with open(os.path.join(d, f"{case}_rows.txt"), "w") as fo:
            fo.write(" ".join(str(int(i)) for i in idx) + "\n")
        for kind in ["envi", "fluxes", "trackers"]:
            lines = open(os.path.join(tmp, f"dbg_{kind}.log")).read().split("\n")
            assert lines[-1] == "" and len(lines) == n + hdr + 1
            keep = lines[:hdr] + [lines[hdr + int(i)] for i in idx]
            with gzip.GzipFile(os.path.join(d, f"{case}_{kind}.log.gz"), "wb", mtime=0) as fo:
                fo.write(("\n".join(keep) + "\n").encode())
        shutil.rmtree(tmp)
        print("debug log", case, n, "rows, kept", len(idx))


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref", "oracle"])
    copy_smoke()
    smoke_records()
    synthetic()
    warning_counts()
    restart_cases()
    balance_case()
    events_infra_case()
    sipnet_infra_case()
    debug_log_cases()
    subprocess.run(["du", "-sh", GOLD])
