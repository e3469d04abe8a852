#!/usr/bin/env python3
"""dev: randomised differential test, GPU throughput / strict kernels against the oracle.
Random valid flag sets, perturbed members (some pushed to mortality / drought), random event
schedules, random segmentation of the run, both math policies.  usage: fuzz_gpu.py [trials] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):                 # an experimental / older build (tools/build_variants.py)
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
from tests import helpers

# FUZZ_COOP=1: a campaign on the cooperative kernels only -- default flags, throughput arithmetic, every
# layout forced in turn, and a few fragile members that single events kill in the middle of a run
COOP_ONLY = bool(os.environ.get("FUZZ_COOP"))
# FUZZ_NCYC=1: a campaign on the nitrogen-cycle flag set (litter pool + anaerobic + nitrogen cycle): the
# cooperative kernel of that set (stepCoopNKernel: carbon / water / light / soil wavefronts) forced or picked
# by the shape policy, fragile members, and productive members short of nitrogen (the exact-supply
# hand-over of checkNitrogenLimitation)
NCYC_ONLY = bool(os.environ.get("FUZZ_NCYC"))
# FUZZ_OPT=1: a campaign on the optional-physics instantiations of the cooperative layouts (stepCoopXKernel & co,
# stepCoopNXKernel & co: run-time flags): always some optional flag on, throughput arithmetic, lean launches, the
# one- and two-chunk layouts forced in turn or picked by the shape policy, regular tiles on and off, fragile stands
OPT_ONLY = bool(os.environ.get("FUZZ_OPT"))
# FUZZ_RAGGED=1 (on top of any of the above, or alone): every trial has several sites and the sites' forcings end at
# different records (a site's members stop at ITS last record; the launch cuts fall before, at and after those ends)
RAGGED = bool(os.environ.get("FUZZ_RAGGED"))
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1      # rerun one trial with details
rng = None
oracle = helpers.load_oracle()
ALLP = os.path.join(helpers.GOLDEN, "synth", "allflags.param")
FERT, HARVEST, IRRIG, PLANT, TILL, LEAFON, LEAFOFF = range(7)

def random_flags():
    f = dict(events=int(rng.random() < 0.8), gdd=1, growthResp=int(rng.random() < 0.3),
             leafWater=int(rng.random() < 0.3), litterPool=int(rng.random() < 0.5),
             waterHResp=1, anaerobic=0, nitrogenCycle=0, flooding=int(rng.random() < 0.3),
             carbonSaturation=0, soilPhenol=0)
    r = rng.random()
    if r < 0.2: f["gdd"] = 0
    elif r < 0.4: f["gdd"], f["soilPhenol"] = 0, 1
    if rng.random() < 0.4: f["anaerobic"] = 1
    elif rng.random() < 0.2: f["waterHResp"] = 0
    if f["anaerobic"] and f["litterPool"] and rng.random() < 0.6: f["nitrogenCycle"] = 1
    if f["litterPool"] and rng.random() < 0.4: f["carbonSaturation"] = 1
    if rng.random() < 0.25 or COOP_ONLY: f = {}          # default flags: throughput / cooperative kernels
    if NCYC_ONLY: f = dict(litterPool=1, anaerobic=1, nitrogenCycle=1, events=int(rng.random() < 0.8))
    if OPT_ONLY:
        while True:
            f = dict(events=int(rng.random() < 0.8), growthResp=int(rng.random() < 0.4), leafWater=int(rng.random() < 0.4),
                     litterPool=int(rng.random() < 0.5), flooding=int(rng.random() < 0.3), anaerobic=int(rng.random() < 0.4),
                     waterHResp=1, carbonSaturation=0, nitrogenCycle=0)
            if not f["anaerobic"] and rng.random() < 0.2: f["waterHResp"] = 0
            if f["litterPool"] and rng.random() < 0.4: f["carbonSaturation"] = 1
            if f["anaerobic"] and f["litterPool"] and rng.random() < 0.4: f["nitrogenCycle"] = 1
            r = rng.random()
            if r < 0.2: f["gdd"] = 0
            elif r < 0.4: f["gdd"], f["soilPhenol"] = 0, 1
            extras = f["growthResp"] + f["leafWater"] + f["flooding"] + f["carbonSaturation"]
            if (f["nitrogenCycle"] and extras) or (not f["nitrogenCycle"] and (extras or f["litterPool"] or f["anaerobic"])):
                break
    return f

def random_events(clim, n):
    ev = []
    ndays = clim.n_steps // 48
    idx = sorted(rng.choice(np.arange(1, ndays), size=min(n, ndays - 1), replace=False))
    for k in idx:
        e = sa.Event(); e.year = int(clim.year[k * 48]); e.day = int(clim.day[k * 48])
        typ = int(rng.integers(0, 7)); e.type = typ
        p = {FERT: (rng.uniform(0, 20), rng.uniform(0, 60), rng.uniform(0, 10)),
             HARVEST: tuple(rng.dirichlet([1, 1, 1, 1, 4])[:4]) if rng.random() < 0.8 else (1.0, 1.0, 0.0, 0.0),
             IRRIG: (rng.uniform(0.1, 5), int(rng.integers(0, 2))),
             PLANT: (rng.uniform(0, 30), rng.uniform(0, 300), rng.uniform(0, 40), rng.uniform(0, 40)),
             TILL: (rng.uniform(0.05, 0.6),), LEAFON: (), LEAFOFF: ()}[typ]
        if typ == HARVEST and len(p) == 4 and p != (1.0, 1.0, 0.0, 0.0):
            a, b2 = rng.random(), rng.random()      # removed + transferred <= 1 above and below ground
            p = (a * 0.9, b2 * 0.9, (1 - a) * 0.9, (1 - b2) * 0.9)
        for i, v in enumerate(p): e.p[i] = float(v)
        ev.append(e)
    return ev

worst = 0.0
t00 = time.time()
for trial in range(trials):
    if only >= 0 and trial != only:
        continue
    rng = np.random.default_rng(seed0 * 100003 + trial)     # every trial reproducible on its own
    kw = random_flags()
    flags = sa.flags_from(**kw)
    base = sa.read_params(ALLP, flags)[0]
    M = int(rng.choice([1, 7, 64, 70, 130, 200]))
    T = 48 * int(rng.integers(20, 200))
    start = int(rng.integers(0, 300)) * 48
    S = int(rng.choice([1, 1, 1, 2, 3, 8]))            # sites (8: the XCD-aware block mapping)
    site0 = int(rng.integers(0, 5))
    site_T = [T] * S
    if RAGGED:      # (a generator of its own: the trials of the other campaigns stay what their seeds made them)
        rng_r = np.random.default_rng(seed0 * 7919 + trial)
        S = int(rng_r.choice([2, 3, 5, 8]))
        site_T = [int(x) for x in rng_r.integers(1, T + 1, size=S)]
        if rng_r.random() < 0.3: site_T[0] = int(rng_r.choice([1, 15, 16, 17, 47]))
        site_T[int(rng_r.integers(1, S))] = T          # the batch is as long as its longest site
    clims = []
    for sidx in range(S):
        raw = synth.half_hourly_year_raw(start + T, site=site0 + sidx)
        raw = {k: v[start:start + site_T[sidx]] for k, v in raw.items()}
        clims.append(synth.convert_raw(synth.round_like_file(raw)))
    clim = clims[int(np.argmax(site_T))]     # (the longest: the event schedule spans it)
    members = synth.perturbed_params(base, M, seed=int(rng.integers(1 << 30)), scale=float(rng.choice([1.0, 3.0])))
    if M > 3:      # a few hard cases
        members[1, pi("plantWoodInit")] *= 0.001        # barely alive
        members[2, pi("soilWFracInit")] = 0.02          # drought
        members[3, pi("leafTurnoverRate")] = 0.9
    ev = random_events(clim, int(rng.integers(1, 12))) if flags[0] else None
    if COOP_ONLY and M > 8:
        for mm in rng.choice(M, size=min(4, M // 8), replace=False):      # fragile stands: a harvest may finish them off
            members[mm, pi("plantWoodInit")] *= float(10.0 ** -rng.uniform(1.5, 4.0))
    if OPT_ONLY and M > 8:
        for mm in rng.choice(M, size=min(4, M // 8), replace=False):      # fragile stands: a harvest may finish them off
            members[mm, pi("plantWoodInit")] *= float(10.0 ** -rng.uniform(1.5, 4.0))
        n_ = M
        members[:, pi("soilCSaturation")] = members[:, pi("soilInit")] * rng.uniform(0.5, 3.0, n_)
        members[:, pi("waterDrainFrac")] = rng.uniform(0.2, 1.0, n_)
        members[:, pi("leafPoolDepth")] *= rng.uniform(0.02, 1.5, n_)
        if rng.random() < 0.3: members[:, pi("anaerobicTransExp")] = rng.uniform(1.0, 3.0, n_)
    if NCYC_ONLY and M > 8:
        for mm in rng.choice(M, size=min(4, M // 8), replace=False):      # fragile stands
            members[mm, pi("plantWoodInit")] *= float(10.0 ** -rng.uniform(1.5, 4.0))
        for mm in rng.choice(M, size=min(6, M // 6), replace=False):      # productive, and short of nitrogen
            members[mm, pi("aMax")] *= float(rng.uniform(2.0, 4.0))
            members[mm, pi("baseVegResp")] *= float(rng.uniform(0.2, 0.5))
            if rng.random() < 0.5:
                members[mm, pi("mineralNInit")] = float(10.0 ** -rng.uniform(1, 5))
                members[mm, pi("plantStorageNInit")] = float(10.0 ** -rng.uniform(0, 4))
                members[mm, pi("soilOrgNInit")] *= float(10.0 ** -rng.uniform(0, 3))
    fast = bool(rng.random() < 0.7) or COOP_ONLY or NCYC_ONLY or OPT_ONLY
    prec = sa.F32_MIXED if (fast and rng.random() < 0.25) else sa.F64
    runs = [oracle.run_block(flags, members, c, ev) for c in clims]
    # (rows past a site's last record: nothing is computed there, neither side is looked at)
    want = np.concatenate([np.concatenate([r[0], np.zeros((3, T - r[0].shape[1], M))], axis=1) for r in runs], axis=2)
    valid = np.concatenate([np.repeat((np.arange(T) < tl)[:, None], M, axis=1) for tl in site_T], axis=1)
    final = np.concatenate([r[1] for r in runs], axis=0)
    st = np.concatenate([r[2] for r in runs])
    # kernel choice: default policy, or forced one-wave / HBM-ring cooperative / run-time flags
    forced, kern, kopt = "", sa.KERNEL_AUTO, 0
    r = rng.random()
    default_flags = not any(v != sa.DEFAULT_FLAGS.get(k) for k, v in kw.items())
    if fast or prec == sa.F32_MIXED:
        if r < 0.3: kern = sa.KERNEL_ONE_WAVE; forced = " one-wave"
        elif r < 0.4 and default_flags: kern = sa.KERNEL_COOP_HBM; forced = " coop-hbm"
        elif r < 0.5 and default_flags: kern = sa.KERNEL_COOP_PAIR; forced = " coop-pair"
        elif r < 0.56 and default_flags: kern = sa.KERNEL_COOP_QUAD; forced = " coop-quad"
        elif r < 0.6 and default_flags: kern = sa.KERNEL_COOP_LDS; forced = " coop-lds"
    if rng.random() < 0.3: kopt = sa.KOPT_RUNTIME_FLAGS; forced += " rt-flags"
    if COOP_ONLY:
        kopt = 0
        kern, forced = [(sa.KERNEL_AUTO, ""), (sa.KERNEL_COOP_LDS, " coop-lds"), (sa.KERNEL_COOP_HBM, " coop-hbm"),
                        (sa.KERNEL_COOP_PAIR, " coop-pair"), (sa.KERNEL_COOP_QUAD, " coop-quad")][int(rng.integers(0, 5))]
    if NCYC_ONLY:
        kopt = sa.KOPT_NO_REGULAR_TILES if rng.random() < 0.2 else 0
        kern, forced = [(sa.KERNEL_AUTO, ""), (sa.KERNEL_COOP_NCYCLE, " coop-ncycle"), (sa.KERNEL_COOP_NCYCLE_PAIR, " coop-ncycle-pair"),
                        (sa.KERNEL_ONE_WAVE, " one-wave")][int(rng.integers(0, 4))]
    if OPT_ONLY:
        kopt = sa.KOPT_NO_REGULAR_TILES if rng.random() < 0.2 else 0
        if flags[sa.FLAG_NAMES.index("nitrogenCycle")]:
            kern, forced = [(sa.KERNEL_AUTO, ""), (sa.KERNEL_COOP_NCYCLE, " coop-ncycle"), (sa.KERNEL_COOP_NCYCLE_PAIR, " coop-ncycle-pair")][int(rng.integers(0, 3))]
        else:
            kern, forced = [(sa.KERNEL_AUTO, ""), (sa.KERNEL_COOP_LDS, " coop-lds"), (sa.KERNEL_COOP_HBM, " coop-hbm"),
                            (sa.KERNEL_COOP_PAIR, " coop-pair")][int(rng.integers(0, 4))]
    # round 5 (a generator of its own: the trials of earlier campaigns stay what their seeds made them): FUZZ_R5=1 adds
    # the instantiations that are new -- optional physics at four chunks per workgroup (fp32-mixed), the record and the
    # diagnostics counters from the optional-physics and nitrogen-cycle kernels ("everything" included)
    r5_full = False
    if os.environ.get("FUZZ_R5") and (OPT_ONLY or NCYC_ONLY):
        rng5 = np.random.default_rng(seed0 * 104729 + trial)
        ncyc_on = bool(flags[sa.FLAG_NAMES.index("nitrogenCycle")])
        if OPT_ONLY and not ncyc_on and prec == sa.F32_MIXED and rng5.random() < 0.5:
            kern, forced = sa.KERNEL_COOP_QUAD, " coop-quad"
        r5_full = bool(rng5.random() < 0.4) and kern not in (sa.KERNEL_COOP_QUAD, sa.KERNEL_COOP_NCYCLE_PAIR)
    if os.environ.get("FUZZ_BOUNDED"): kopt |= sa.KOPT_BOUNDED_WAITS      # the cooperative kernels' build with bounded waits: a protocol bug ends the launch with a report instead of hanging the GPU
    if os.environ.get("FUZZ_KOPT"): kopt = int(os.environ["FUZZ_KOPT"])
    if os.environ.get("FUZZ_KERNEL"):     # rerun a trial on another kernel (with the trial index as third argument)
        kern = getattr(sa, "KERNEL_" + os.environ["FUZZ_KERNEL"].upper()); forced = " forced-" + os.environ["FUZZ_KERNEL"]
    b = sa.Batch(flags, S, M, prec, fast_math=fast, kernel=kern, kernel_options=kopt)
    for sidx in range(S):
        if ev is not None: b.set_events(sidx, ev)
        b.set_climate(sidx, clims[sidx]); b.set_params(sidx, members)
    # a third of the trials also ask for the 44-column record and the diagnostics counters (the
    # "full" instantiations of the throughput kernels, or the strict kernel's)
    want_full = bool(rng.random() < 0.33) and kern not in (sa.KERNEL_COOP_QUAD, sa.KERNEL_COOP_NCYCLE, sa.KERNEL_COOP_NCYCLE_PAIR)    # no full-state builds of these
    if OPT_ONLY: want_full = False       # (round 4's campaigns: lean only)
    if r5_full:
        want_full = True
        kopt &= ~sa.KOPT_BOUNDED_WAITS   # (the bounded-wait build has lean instantiations only)
    if want_full:
        b.enable_diagnostics()
        forced += " full"
    b.setup()
    cuts = sorted(set([0, T] + [int(x) for x in rng.integers(1, T, size=int(rng.integers(0, 4)))]))
    runs_g = [b.run(a0, a1 - a0, full=want_full) for a0, a1 in zip(cuts[:-1], cuts[1:])]
    got = torch.cat([r_[0] for r_ in runs_g], dim=1).double().cpu().numpy()
    got = np.where(valid[None], got, 0.0)
    if want_full:
        rec_g = torch.cat([r_[1] for r_ in runs_g], dim=0).cpu().numpy()
        diag_g = b.get_diagnostics()
    b_kernel = b.last_launch()["kernel"]
    status = np.asarray(b.get_status()); state = b.get_state()
    sums_ok = bool(os.environ.get("FUZZ_SUMS")) and not want_full and b.sums_in_kernel()
    b.close()
    if sums_ok:
        # round 6 (FUZZ_SUMS=1; a generator of its own): the same trial through sipnet_batch_run_sums -- every member's sums over
        # groups of k steps out of the step kernel's launch, launches cut at random multiples of k -- against the planes just
        # computed, added up on the host in step order: bit for bit for every member that runs
        rs = np.random.default_rng(1000003 * int(trial) + 17)
        k = int(rs.choice([1, 2, 7, 16, 48, 100]))
        G = (T + k - 1) // k
        scuts = sorted(set([0, T] + [int(x) * k for x in rs.integers(1, max(G, 2), size=int(rs.integers(0, 3))) if int(x) * k < T]))
        bs = sa.Batch(flags, S, M, prec, fast_math=fast, kernel=kern, kernel_options=kopt & ~sa.KOPT_BOUNDED_WAITS)
        for sidx in range(S):
            if ev is not None: bs.set_events(sidx, ev)
            bs.set_climate(sidx, clims[sidx]); bs.set_params(sidx, members)
        bs.setup()
        parts = [bs.run_sums(a0, a1 - a0, k).cpu().numpy() for a0, a1 in zip(scuts[:-1], scuts[1:])]
        ks = bs.last_launch()["kernel"]
        st_s = bs.get_state(); bs.close()
        got_s = np.concatenate(parts, axis=1)
        planes_raw = torch.cat([r_[0] for r_ in runs_g], dim=1).double().cpu().numpy()
        want_s = np.zeros_like(got_s)
        for t_ in range(T):
            want_s[:, t_ // k] += np.where(valid[t_][None], planes_raw[:, t_], 0.0)
        okc = (np.asarray(status) == 0)
        # (groups a shorter site's launch does not reach are not written: compare the groups that hold a valid step)
        gvalid = np.zeros((G, valid.shape[1]), dtype=bool)
        for t_ in range(T):
            gvalid[t_ // k] |= valid[t_]
        sel = gvalid[None] & okc[None, None, :]
        # (the one-wavefront sums build -- compiled, like the plain one, with -ffp-contract=fast -- fuses a few products on rare paths
        # differently from the build that stores the planes: a flux there differs in its last bit once in ~10^6 values, so its sums
        # equal the planes' to the arithmetic's last bits, not bit for bit; every cooperative sums kernel is held to bits)
        loose = ks.startswith("stepFastSumsKernel<")
        tol_ = (1e-13 if "<double" in ks else 1e-6) * np.abs(np.where(sel, want_s, 0.0)).max() if loose else 0.0
        if not (np.abs(np.where(sel, got_s - want_s, 0.0)) <= tol_).all():
            d_ = np.abs(np.where(sel, got_s - want_s, 0.0)); i_ = np.unravel_index(d_.argmax(), d_.shape)
            raise AssertionError(f"MISMATCH (sums) trial {trial}: k {k} cuts {scuts} kernel {ks}: {d_.max():.3e} at plane/group/column {i_}")
        if loose:
            assert np.allclose(st_s[okc], state[okc], rtol=1e-11 if "<double" in ks else 1e-5, atol=1e-300 if "<double" in ks else 1e-7), \
                f"MISMATCH (state after sums, one-wave) trial {trial}"
        else:
            assert np.array_equal(st_s[okc], state[okc]), f"MISMATCH (state after sums) trial {trial}"
        forced += f" sums(k={k},{len(scuts) - 1} launches,{ks.split('<')[0]})"
    if only >= 0 and os.environ.get("FUZZ_STOP"):    # pools of one member after N steps, this kernel vs the one-wave kernel
        nstop, mdbg = int(os.environ["FUZZ_STOP"]), int(os.environ.get("FUZZ_MEMBER", "0"))
        for kk, nm in ((kern, "forced"), (sa.KERNEL_ONE_WAVE, "one-wave")):
            bb = sa.Batch(flags, S, M, prec, fast_math=fast, kernel=kk, kernel_options=kopt)
            for sidx in range(S):
                if ev is not None: bb.set_events(sidx, ev)
                bb.set_climate(sidx, clims[sidx]); bb.set_params(sidx, members)
            bb.setup()
            for a0, a1 in zip(cuts[:-1], cuts[1:]):
                if a0 >= nstop: break
                bb.run(a0, min(a1, nstop) - a0)
            st_ = bb.get_state()[mdbg]
            print("           %-8s after %d steps, member %d:" % (nm, nstop, mdbg), " ".join("%.10g" % v for v in st_[:31]))
            bb.close()
    if want_full:
        # two members of the first site against the oracle's records and counters
        for m in sorted(set([0, M - 1])):
            if st[m] != 0:
                continue
            so, rec_o, dg = oracle.run_member(flags, members[m], clims[0], ev)
            assert so == 0
            cs = np.maximum(np.abs(rec_o).max(axis=0), 1e-3)
            rtol = 1e-9 if prec == sa.F64 else 5e-3
            rel_ = np.abs(rec_g[:site_T[0], :36, m] - rec_o) / cs
            rerr = rel_.max()
            if prec == sa.F32_MIXED and rerr < 5e-2 and float((rel_ > rtol).mean()) < 1e-4:
                rerr = 0.0      # fp32 flux arithmetic: a threshold branch taken a step apart, judged by share (as the planes are)
            if not rerr < rtol:
                t_, c_ = np.unravel_index(rel_.argmax(), rel_.shape)
                raise AssertionError(f"MISMATCH (record) trial {trial} member {m}: {rerr:.3e} at step {t_} column {c_} "
                                     f"(gpu {rec_g[t_, c_, m]!r} oracle {rec_o[t_, c_]!r}) kernel {b_kernel}{forced} flags {kw}")
            if prec == sa.F64:
                assert diag_g["n_clamp_warn"][m] == dg.n_clamp_warn, ("clamp warnings", m, diag_g["n_clamp_warn"][m], dg.n_clamp_warn)
                assert diag_g["n_balance_warn"][m] == dg.n_balance_warn, ("balance warnings", m)
    ok = (st == 0)
    assert (status[ok] == 0).all() and ((status != 0) == (st != 0)).all(), (trial, status, st)
    # error relative to each plane's maximum, with an absolute floor (a plane can be all ~0:
    # winter GPP) of 1e-3 gC m-2 (or cm) per step
    scale = np.maximum(np.abs(want[:, :, ok]).max(axis=(1, 2), keepdims=True), 1e-3)
    err = (np.abs(got[:, :, ok] - want[:, :, ok]) / scale).max() if ok.any() else 0.0
    pfloor = 1.0 if prec == sa.F32_MIXED else 1e-2    # fp32 fluxes leave ~1e-5 gC residues in emptied pools
    perr = (np.abs(state[ok, :13] - final[ok, 14:27]) / np.maximum(np.abs(final[ok, 14:27]), pfloor)).max() if ok.any() else 0.0
    tol = 1e-9
    ptol = 5e-3 if prec == sa.F32_MIXED else 1e-8
    if prec == sa.F32_MIXED and ok.any():
        # fp32 flux arithmetic can take a threshold branch (snow gone, soil dry) one step away
        # from fp64; judge it on the share of outliers and on the time sums instead of the maximum
        rel = np.abs(got[:, :, ok] - want[:, :, ok]) / scale
        outliers = float((rel > 1e-4).mean())
        sums = np.abs(got[:, :, ok].sum(1) - want[:, :, ok].sum(1)) / (np.abs(want[:, :, ok]).sum(1) + 1.0)
        print(f"           fp32: max {err:.2e}, share of values off by > 1e-4 of the plane maximum {outliers:.2e}, "
              f"worst time-sum error {sums.max():.2e}")
        if only >= 0:
            pl, stp, mem = np.unravel_index(rel.argmax(), rel.shape)
            share = (rel > 1e-4).mean(axis=(0, 1))
            print("           worst plane/step/member", pl, stp, mem, "got", got[:, :, ok][pl, stp, mem], "want", want[:, :, ok][pl, stp, mem])
            print("           share of outliers per member:", np.round(share, 3))
            bad = int(share.argmax())
            first = int(np.nonzero((rel[:, :, bad] > 1e-4).any(axis=0))[0][0])
            print("           member", bad, "first outlier step", first, "day", clim.day[first], "got", got[:, first, bad], "want", want[:, first, bad])
        assert outliers < 2e-3 and sums.max() < 2e-3, "MISMATCH (fp32)"
        err = 0.0
    flag_s = "+".join(k for k, v in kw.items() if v != sa.DEFAULT_FLAGS.get(k)) or "default"
    print(f"trial {trial:3d}: S={S} M={M:3d} T={T:5d} segs={len(cuts)-1} {'f32' if prec else 'f64'} {'fast' if fast else 'strict'} "
          f"ev={0 if ev is None else len(ev):2d} [{flag_s}]{forced} {b_kernel.split('<')[0]} planes {err:.2e} pools {perr:.2e}", flush=True)
    assert np.isfinite(got[:, :, ok]).all()
    if only >= 0 and prec == sa.F64 and ok.any():   # where a mismatch starts
        rel = np.abs(got[:, :, ok] - want[:, :, ok]) / scale
        badm = np.nonzero((rel > 1e-9).any(axis=(0, 1)))[0]
        print("           cuts", cuts, "kernel", b_kernel, "members over 1e-9:", badm[:16], "of", int(ok.sum()))
        if len(badm):
            m0 = int(badm[0]); first = int(np.nonzero((rel[:, :, m0] > 1e-9).any(axis=0))[0][0])
            print("           member", m0, "first step", first, "year/day", clim.year[first], clim.day[first],
                  "got", got[:, first, ok][:, m0], "want", want[:, first, ok][:, m0])
            steps_bad = np.nonzero((rel[:, :, m0] > 1e-9).any(axis=0))[0]
            print("           member", m0, "steps over 1e-9:", steps_bad[:40], "count", len(steps_bad))
            print("           at that step, all members' |dNEE|/scale:", np.round(rel[0, first, :] * 1e6, 2)[:16], "(x1e-6)")
            print("           NEE got/want around:", [(int(tt), float(got[0, tt, ok][m0]), float(want[0, tt, ok][m0])) for tt in range(first - 2, first + 20)])
            if os.environ.get("FUZZ_REC"):   # the 44-column record of that member around that step, kernel vs oracle
                bb = sa.Batch(flags, S, M, prec, fast_math=fast, kernel=kern, kernel_options=kopt)
                for sidx in range(S):
                    if ev is not None: bb.set_events(sidx, ev)
                    bb.set_climate(sidx, clims[sidx]); bb.set_params(sidx, members)
                bb.setup()
                recs_ = [bb.run(a0, a1 - a0, full=True)[1] for a0, a1 in zip(cuts[:-1], cuts[1:])]
                rec_g = torch.cat(recs_, dim=0).cpu().numpy()
                bb.close()
                _, rec_o, _ = oracle.run_member(flags, members[m0], clims[0], ev)
                for tt in (first - 1, first, first + 1):
                    d_ = np.abs(rec_g[tt, :36, m0] - rec_o[tt])
                    print("           step", tt, "record columns differing > 1e-12:", {int(c): (float(rec_g[tt, c, m0]), float(rec_o[tt, c])) for c in np.nonzero(d_ > 1e-12)[0]})
            if ev is not None:
                print("           events:", [(e.year, e.day, e.type, [round(x, 3) for x in e.p[:4]]) for e in ev])
    assert err < tol and perr < ptol, "MISMATCH"
    worst = max(worst, err if prec == sa.F64 else 0.0)
print(f"{trials} trials ok in {time.time()-t00:.0f} s; worst fp64 plane error {worst:.2e} of the plane maximum")
