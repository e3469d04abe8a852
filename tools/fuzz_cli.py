#!/usr/bin/env python3
"""dev: process-level differential test -- the drop-in CLI (GPU) against the REAL reference binary
(oracle/_ref/sipnet_ref, which travels to the GPU box) on randomised run directories: random valid
flag sets in sipnet.in, perturbed parameter files, random event files, random output options.
Compared: exit codes, sipnet.config (byte-identical below its time-stamp line), sipnet.out and
events.out (same lines and tokens; numbers within half a unit of the last printed digit + 1e-6
relative, since OCML's pow / exp differ from glibc's in the last bits), single-variable outputs.
usage: fuzz_cli.py [trials] [seed]"""
import gzip, os, shutil, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

REF_BIN = os.path.join(REPO, "oracle", "_ref", "sipnet_ref")
CLI = os.path.join(REPO, "sipnet_amd", "bin", "sipnet")
G = os.path.join(REPO, "tests", "golden", "smoke")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if not os.path.exists(REF_BIN):
    print("oracle/_ref/sipnet_ref is not on this box"); sys.exit(3)

def random_flags(rng):
    f = dict(EVENTS=int(rng.random() < 0.8), GDD=1, GROWTH_RESP=int(rng.random() < 0.3),
             LEAF_WATER=int(rng.random() < 0.3), LITTER_POOL=int(rng.random() < 0.5), WATER_HRESP=1,
             ANAEROBIC=0, NITROGEN_CYCLE=0, FLOODING=int(rng.random() < 0.3), CARBON_SATURATION=0,
             SOIL_PHENOL=0, SNOW=int(rng.random() < 0.9))
    r = rng.random()
    if r < 0.2: f["GDD"] = 0
    elif r < 0.4: f["GDD"], f["SOIL_PHENOL"] = 0, 1
    if rng.random() < 0.4: f["ANAEROBIC"] = 1
    elif rng.random() < 0.2: f["WATER_HRESP"] = 0
    if f["ANAEROBIC"] and f["LITTER_POOL"] and rng.random() < 0.6: f["NITROGEN_CYCLE"] = 1
    if f["LITTER_POOL"] and rng.random() < 0.4: f["CARBON_SATURATION"] = 1
    return f

def write_case(d, rng):
    flags = random_flags(rng)
    # user-specified leaf events exclude every computed phenology (the reference exits 3 otherwise;
    # a tenth of the trials keep the conflict to check that both binaries reject it alike)
    user_leaf = rng.random() < 0.3
    conflict = rng.random() < 0.1
    if user_leaf and not conflict:
        flags["GDD"], flags["SOIL_PHENOL"] = 0, 0
    opts = dict(DO_MAIN_OUTPUT=1, DO_SINGLE_OUTPUTS=int(rng.random() < 0.3), DUMP_CONFIG=1,
                PRINT_HEADER=int(rng.random() < 0.5), QUIET=1)
    with open(os.path.join(d, "sipnet.in"), "w") as f:
        for k, v in {**opts, **flags}.items():
            f.write(f"{k} = {v}\n")
    keep = {"leafAllocation", "woodAllocation", "fineRootAllocation", "fineRootFrac", "coarseRootFrac",
            "laiInit", "leafOnDay", "leafOffDay", "dVpdExp", "soilRespMoistEffect", "cFracLeaf"}
    out = []
    for line in open(os.path.join(G, "russell_2", "sipnet.param")):
        t = line.split()
        bounded = any(k in t[0] for k in ("Frac", "Allocation", "Eff", "fAnoxia")) if t else True
        if len(t) >= 2 and t[0] not in keep and not bounded and not t[0].startswith("!") and rng.random() < 0.5:
            t[1] = repr(float(t[1]) * float(rng.uniform(0.85, 1.15)))
        if len(t) >= 2 and t[0] in ("leafOnDay", "leafOffDay") and user_leaf and not conflict:
            t[1] = "0"
        out.append(" ".join(t))
    extra = {"soilCSaturation": 3000.0 * rng.uniform(0.5, 2), "waterDrainFrac": rng.uniform(0.2, 1.0),
             "soilTempLeafOn": rng.uniform(2, 12), "growthRespFrac": 0.2, "leafPoolDepth": 0.1}
    have = {l.split()[0] for l in out if l.split()}
    out += [f"{k} {v!r}" for k, v in extra.items() if k not in have]
    open(os.path.join(d, "sipnet.param"), "w").write("\n".join(out) + "\n")
    # forcing: russell (3-hourly, two years) or niwot (legacy 14-column file with a location
    # column, two steps of unequal length per day: the irregular running-mean ring schedule)
    forcing = "niwot" if rng.random() < 0.35 else "russell_1"
    with gzip.open(os.path.join(G, forcing, "sipnet.clim.gz"), "rb") as fi, open(os.path.join(d, "sipnet.clim"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    n = int(rng.integers(0, 14))
    if forcing == "niwot":
        days, year = sorted(set(int(x) for x in rng.integers(2, 300, size=n))), 1999
    else:
        days, year = sorted(set(int(x) for x in rng.integers(2, 360, size=n))), 2016
    with open(os.path.join(d, "events.in"), "w") as f:
        for day in days:
            typ = ["fert", "harv", "irrig", "plant", "till", "leafon", "leafoff"][int(rng.integers(0, 7 if user_leaf else 5))]
            if typ == "fert": p = f"{rng.uniform(0, 20):.3f} {rng.uniform(0, 60):.3f} {rng.uniform(0, 10):.3f}"
            elif typ == "harv":
                a, b = rng.random(), rng.random()
                p = "1 1 0 0" if rng.random() < 0.15 else f"{a*0.9:.3f} {b*0.9:.3f} {(1-a)*0.9:.3f} {(1-b)*0.9:.3f}"
            elif typ == "irrig": p = f"{rng.uniform(0.1, 5):.2f} {int(rng.integers(0, 2))}"
            elif typ == "plant": p = f"{rng.uniform(0, 30):.2f} {rng.uniform(0, 300):.2f} {rng.uniform(0, 40):.2f} {rng.uniform(0, 40):.2f}"
            elif typ == "till": p = f"{rng.uniform(0.05, 0.6):.3f}"
            else: p = ""
            f.write(f"{year} {day} {typ} {p}\n".replace("  ", " "))
    return flags, opts, n

def isnum(x):
    try: float(x); return True
    except ValueError: return False

def compare_text(a, b, what):
    la, lb = a.split("\n"), b.split("\n")
    assert len(la) == len(lb), f"{what}: {len(la)} vs {len(lb)} lines"
    same = 0
    for i, (x, y) in enumerate(zip(la, lb)):
        if x == y: same += 1; continue
        tx, ty = x.replace(",", " ").replace("=", " ").split(), y.replace(",", " ").replace("=", " ").split()
        assert len(tx) == len(ty), f"{what} line {i}: {x!r} vs {y!r}"
        for u, v in zip(tx, ty):
            if u == v: continue
            assert isnum(u) and isnum(v), f"{what} line {i}: {u!r} vs {v!r}"
            dec = len(v.split(".")[1]) if "." in v else 0
            tol = 0.6 * 10 ** (-dec) + 1e-6 * abs(float(v))
            assert abs(float(u) - float(v)) <= tol, f"{what} line {i}: {u} vs {v}\n{x}\n{y}"
    return same, len(la)

for trial in range(trials):
    rng = np.random.default_rng(seed0 * 100003 + trial)
    da, db = tempfile.mkdtemp(prefix="fz_ref_"), tempfile.mkdtemp(prefix="fz_gpu_")
    flags, opts, nev = write_case(da, rng)
    for f in os.listdir(da): shutil.copyfile(os.path.join(da, f), os.path.join(db, f))
    ra = subprocess.run([REF_BIN, "-i", "sipnet.in"], cwd=da, capture_output=True, text=True)
    rb = subprocess.run([CLI, "-i", "sipnet.in"], cwd=db, capture_output=True, text=True)
    tag = "+".join(k for k, v in flags.items() if v != dict(EVENTS=1, GDD=1, WATER_HRESP=1, SNOW=1).get(k, 0))
    assert ra.returncode == rb.returncode, (trial, ra.returncode, rb.returncode, ra.stdout[-500:], rb.stdout[-500:])
    if ra.returncode != 0:
        why = [l for l in ra.stdout.split("\n") if "ERROR" in l]
        print("           reference:", why[-1][:150] if why else ra.stdout[-150:])
    report = []
    if ra.returncode == 0:
        body = lambda p: open(p).read().split("\n", 1)[1]
        assert body(os.path.join(da, "sipnet.config")) == body(os.path.join(db, "sipnet.config")), "sipnet.config differs"
        names = ["sipnet.out"] + (["events.out"] if flags["EVENTS"] else [])
        if opts["DO_SINGLE_OUTPUTS"]: names += ["sipnet.NEE", "sipnet.NEE_cum", "sipnet.GPP", "sipnet.GPP_cum"]
        for nm in names:
            same, tot = compare_text(open(os.path.join(db, nm)).read(), open(os.path.join(da, nm)).read(), nm)
            report.append(f"{nm} {same}/{tot}")
        assert os.path.exists(os.path.join(da, "events.out")) == os.path.exists(os.path.join(db, "events.out"))
    # restart interchange on a random midnight: OUR segment 1 + the REFERENCE resuming from our
    # checkpoint, and the reference's segment 1 + US resuming from its checkpoint, must both
    # reproduce the reference's continuous run (restart.c; sipnet.c:1963-1989)
    if ra.returncode == 0 and rng.random() < 0.5:
        lines = open(os.path.join(da, "sipnet.clim")).read().split("\n")[:-1]
        c0 = 1 if len(lines[0].split()) == 14 else 0       # legacy files start with a location column
        days = [i for i in range(1, len(lines)) if lines[i].split()[c0:c0 + 2] != lines[i - 1].split()[c0:c0 + 2]]
        k = days[int(rng.integers(len(days) // 5, 4 * len(days) // 5))]
        by, bd = int(lines[k].split()[c0]), int(lines[k].split()[c0 + 1])
        evl = open(os.path.join(da, "events.in")).read().split("\n")[:-1]
        before = [l for l in evl if (int(l.split()[0]), int(l.split()[1])) < (by, bd)]
        after = [l for l in evl if (int(l.split()[0]), int(l.split()[1])) >= (by, bd)]
        full_out = open(os.path.join(da, "sipnet.out")).read()
        hdr = 1 if opts["PRINT_HEADER"] else 0
        for first, second, who in ((CLI, REF_BIN, "ours->ref"), (REF_BIN, CLI, "ref->ours")):
            d1, d2 = tempfile.mkdtemp(prefix="fz_s1_"), tempfile.mkdtemp(prefix="fz_s2_")
            for d_, cl, ev in ((d1, lines[:k], before), (d2, lines[k:], after)):
                for f in ("sipnet.in", "sipnet.param"):
                    shutil.copyfile(os.path.join(da, f), os.path.join(d_, f))
                open(os.path.join(d_, "sipnet.clim"), "w").write("\n".join(cl) + "\n")
                open(os.path.join(d_, "events.in"), "w").write("".join(x + "\n" for x in ev))
            r1 = subprocess.run([first, "-i", "sipnet.in", "--restart-out", "ck"], cwd=d1, capture_output=True, text=True)
            assert r1.returncode == 0, (who, "segment 1", r1.stdout[-800:])
            shutil.copyfile(os.path.join(d1, "ck"), os.path.join(d2, "ck"))
            r2 = subprocess.run([second, "-i", "sipnet.in", "--restart-in", "ck"], cwd=d2, capture_output=True, text=True)
            assert r2.returncode == 0, (who, "segment 2", r2.stdout[-800:])
            o1 = open(os.path.join(d1, "sipnet.out")).read().split("\n")[:-1]
            o2 = open(os.path.join(d2, "sipnet.out")).read().split("\n")[:-1]
            joined = "\n".join(o1 + o2[hdr:]) + "\n"
            same, tot = compare_text(joined, full_out, f"restart {who} sipnet.out")
            report.append(f"restart {who} {same}/{tot}")
            shutil.rmtree(d1); shutil.rmtree(d2)
    # --debug-log: 13 pools, 56 fluxes, 37 tracker fields per step (%.15g: compared to 1e-9
    # relative with a 1e-9 absolute floor -- plantCAccountingDelta is a running sum of differences)
    if ra.returncode == 0 and rng.random() < 0.2:
        for d_, binary in ((da, REF_BIN), (db, CLI)):
            rr = subprocess.run([binary, "-i", "sipnet.in", "--debug-log", "dbg"], cwd=d_, capture_output=True, text=True)
            assert rr.returncode == 0, rr.stdout[-500:]
        worst = 0.0
        for kind in ("envi", "fluxes", "trackers"):
            la = open(os.path.join(da, f"dbg_{kind}.log")).read().split("\n")
            lb = open(os.path.join(db, f"dbg_{kind}.log")).read().split("\n")
            assert len(la) == len(lb), f"debug {kind}: line counts"
            for i, (x, y) in enumerate(zip(la, lb)):
                if x == y: continue
                tx, ty = x.split(), y.split()
                assert len(tx) == len(ty), f"debug {kind} line {i}"
                for u, v in zip(tx, ty):
                    if u == v: continue
                    e = abs(float(u) - float(v)) / max(abs(float(u)), 1.0)
                    worst = max(worst, e)
                    assert e < 1e-9, f"debug {kind} line {i}: {u} vs {v}"
        report.append(f"debug logs worst {worst:.1e}")
    # the ensemble extension: every row of a parameter table is one member of ONE batch; each
    # member's files must equal what the reference writes for that parameter set on its own
    if ra.returncode == 0 and rng.random() < 0.25:
        names = [str(x) for x in rng.choice(["aMax", "psnTOpt", "baseVegResp", "soilWHC", "wueConst", "halfSatPar"], size=3, replace=False)]
        cur = {l.split()[0]: float(l.split()[1]) for l in open(os.path.join(da, "sipnet.param")) if len(l.split()) >= 2 and l.split()[0] in names}
        M = int(rng.integers(2, 6))
        rows = [[cur[n] * float(rng.uniform(0.9, 1.1)) for n in names] for _ in range(M)]
        with open(os.path.join(db, "members.txt"), "w") as f:
            f.write(" ".join(names) + "\n")
            for r_ in rows: f.write(" ".join(repr(v) for v in r_) + "\n")
        re_ = subprocess.run([CLI, "-i", "sipnet.in", "--ensemble-params", "members.txt"], cwd=db, capture_output=True, text=True)
        assert re_.returncode == 0, re_.stdout[-800:]
        same_all = 0
        for m, r_ in enumerate(rows):
            dm = tempfile.mkdtemp(prefix="fz_mem_")
            for f in ("sipnet.in", "sipnet.clim", "events.in"): shutil.copyfile(os.path.join(da, f), os.path.join(dm, f))
            out = []
            for l in open(os.path.join(da, "sipnet.param")):
                t = l.split()
                if len(t) >= 2 and t[0] in names: t[1] = repr(r_[names.index(t[0])])
                out.append(" ".join(t))
            open(os.path.join(dm, "sipnet.param"), "w").write("\n".join(out) + "\n")
            rm = subprocess.run([REF_BIN, "-i", "sipnet.in"], cwd=dm, capture_output=True, text=True)
            assert rm.returncode == 0
            same, tot = compare_text(open(os.path.join(db, f"sipnet.{m}.out")).read(), open(os.path.join(dm, "sipnet.out")).read(), f"member {m} sipnet.out")
            same_all += same == tot
            if flags["EVENTS"]:
                compare_text(open(os.path.join(db, f"events.{m}.out")).read(), open(os.path.join(dm, "events.out")).read(), f"member {m} events.out")
            shutil.rmtree(dm)
        report.append(f"ensemble {same_all}/{M} members identical")
    print(f"trial {trial:3d}: rc={ra.returncode} events={nev:2d} [{tag or 'default'}] identical lines: " + ", ".join(report), flush=True)
    shutil.rmtree(da); shutil.rmtree(db)
print(f"{trials} trials ok")
