"""GPU parity of the throughput kernel's run-time-flag instantiation (step_fast.hip,
Generic = true): every optional model flag of the reference (context.c:35-53 -- litter pool,
nitrogen cycle, anaerobic / methane, carbon saturation, flooding, growth respiration, leaf
water, soil-temperature and calendar phenology, no moisture effect on heterotrophic
respiration) through the lean launch that bench-style callers use, against the oracle on the
same members, climate and events.  Tolerance: 1e-9 absolute on the flux planes
(gC or cm per step; the reference's own tolerance is 1e-6, tUtils.h:57-59), 1e-9 relative on
the final pools."""
import os

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.GOLDEN, "synth", "allflags.param")  # every optional parameter set
POOLS = slice(14, 27)  # record columns holding the 13 pools (include/sipnet_amd.h)


def lean_run(flags, clim, members, events=None, fast=True, prec=sa.F64, kernel=sa.KERNEL_AUTO):
    b = sa.Batch(flags, 1, members.shape[0], prec, fast_math=fast if prec == sa.F64 else None, kernel=kernel)
    if events is not None:
        b.set_events(0, events)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    planes, _ = b.run()
    status = b.get_status()
    name = b.last_launch()["kernel"]
    state = b.get_state()
    got = planes.cpu().numpy()
    b.close()
    return got, state, status, name


def compare(tag, got, state, want, final):
    d = np.abs(got - want).max()
    pools_g = state[:, :13]
    pools_o = final[:, POOLS]
    scale = np.maximum(np.abs(pools_o), 1e-3)
    dp = (np.abs(pools_g - pools_o) / scale).max()
    print(f"{tag}: max|d planes| {d:.3e}  max rel d pools {dp:.3e}")
    per_pool = (np.abs(pools_g - pools_o) / scale).max(axis=0)     # Envi order, state.h:416-463
    print("   per pool:", " ".join(f"{v:.1e}" for v in per_pool), " worst member", int((np.abs(pools_g - pools_o) / scale).max(axis=1).argmax()))
    assert np.isfinite(got).all()
    assert d < 1e-9
    assert dp < 1e-9


@pytest.mark.parametrize("case_name", ["russell_2", "russell_3"])
def test_smoke_cases_with_optional_flags(case_name, oracle, tmp_path):
    """russell_2 = litter pool + nitrogen cycle + anaerobic with the full event file;
    russell_3 = growth respiration + leaf water + litter pool, no moisture effect."""
    case = helpers.load_smoke_case(case_name, str(tmp_path))
    names = {"aMax", "soilWHC", "wueConst", "baseVegResp", "leafCN", "kCN", "litterBreakdownRate"}
    members = synth.perturbed_params(case["params"], 70, names=names)
    got, state, status, _ = lean_run(case["flags"], case["clim"], members, case["events"])
    want, final, st = oracle.run_block(case["flags"], members, case["clim"], case["events"])
    assert (st == 0).all() and (np.asarray(status) == 0).all()
    compare(case_name, got, state, want, final)


def _events_all_types(clim):
    ev = []

    def add(day, typ, *p):
        e = sa.Event()
        e.type = typ
        e.year = int(clim.year[0])
        e.day = day
        for i, v in enumerate(p):
            e.p[i] = v
        ev.append(e)

    FERT, HARVEST, IRRIG, PLANT, TILL, LEAFON, LEAFOFF = range(7)  # include/sipnet_amd.h:91-97
    add(40, FERT, 12.0, 40.0, 6.0)                # org-N, org-C, min-N
    add(45, TILL, 0.3)
    add(120, IRRIG, 2.5, 0)                       # canopy irrigation
    add(121, IRRIG, 1.5, 1)                       # soil irrigation
    add(130, LEAFON)
    add(200, HARVEST, 0.2, 0.05, 0.3, 0.1)
    add(205, PLANT, 10.0, 30.0, 5.0, 8.0)
    add(206, FERT, 0.0, 0.0, 20.0)
    add(300, LEAFOFF)
    return ev


_flags = sa.flags_from
F_EVENTS = sa.FLAG_NAMES.index("events")

FLAG_SETS = {
    "growth_resp": dict(growthResp=1),
    "leaf_water": dict(leafWater=1),
    "litter_pool": dict(litterPool=1),
    "no_water_hresp": dict(waterHResp=0),
    "anaerobic": dict(anaerobic=1),
    "anaerobic_litter": dict(anaerobic=1, litterPool=1),
    "flooding": dict(flooding=1),
    "carbon_saturation": dict(litterPool=1, carbonSaturation=1),
    "soil_phenol": dict(gdd=0, soilPhenol=1),
    "calendar_phenology": dict(gdd=0),
    "no_events": dict(events=0),
    "nitrogen": dict(litterPool=1, anaerobic=1, nitrogenCycle=1),
    "everything": dict(litterPool=1, anaerobic=1, nitrogenCycle=1, carbonSaturation=1,
                       flooding=1, growthResp=1, leafWater=1),
}
ALL_ON = dict(litterPool=1, anaerobic=1, nitrogenCycle=1, carbonSaturation=1, flooding=1,
              growthResp=1, leafWater=1, soilPhenol=1)


@pytest.fixture(scope="module")
def clim60():
    """one synthetic half-hourly year (SURVEY 8d generator): winter snow, leaf-on by GDD / soil
    temperature / calendar, summer drought stress, leaf-off"""
    return synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(17520)))


@pytest.fixture(scope="module")
def members70():
    """70 members (one full chunk + a ragged one): the throughput benchmark's perturbation plus
    log-normal spread on the parameters only the optional flags read"""
    base = sa.read_params(BASE, _flags(**ALL_ON))[0]
    out = synth.perturbed_params(base, 70)
    rng = np.random.default_rng(77)
    for name in ["leafCN", "woodCN", "fineRootCN", "kCN", "litterBreakdownRate",
                 "fracLitterRespired", "fAnoxia", "mineralNInit", "soilOrgNInit", "litterOrgNInit",
                 "plantStorageNInit", "nVolatilizationFrac", "nLeachingFrac",
                 "halfNFixationMax", "growthRespFrac", "leafPoolDepth", "waterDrainFrac",
                 "soilCSaturation", "soilMethaneRate", "litterMethaneRate",
                 "anaerobicTransExp", "leafNResorptionFrac"]:
        k = param_index(name)
        assert k >= 0, name
        f = np.exp(rng.normal(0.0, 0.15, 70))
        f[0] = 1.0
        v = out[:, k] * f
        if name in ("fracLitterRespired", "fAnoxia", "leafNResorptionFrac", "nVolatilizationFrac",
                    "nLeachingFrac"):
            v = np.clip(v, 0.0, 0.95)
        out[:, k] = v
    # carbon saturation level around the soil carbon stock, so that the unit clip is exercised on
    # both sides; partial flood drainage
    out[:, param_index("soilCSaturation")] = out[:, param_index("soilInit")] * rng.uniform(0.5, 3.0, 70)
    out[:, param_index("waterDrainFrac")] = rng.uniform(0.2, 1.0, 70)
    # the synthetic soil temperature peaks near 10 C: thresholds the members cross on different days
    out[:, param_index("soilTempLeafOn")] = rng.uniform(4.0, 9.0, 70)
    return out


@pytest.mark.parametrize("name", list(FLAG_SETS))
def test_each_flag_set_matches_oracle(name, oracle, clim60, members70):
    flags = _flags(**FLAG_SETS[name])
    ev = _events_all_types(clim60) if flags[F_EVENTS] else None
    got, state, status, kernel = lean_run(flags, clim60, members70, ev)
    want, final, st = oracle.run_block(flags, members70, clim60, ev)
    assert (st == 0).all() and (np.asarray(status) == 0).all()
    # AUTO at two chunks: a cooperative kernel for EVERY flag set -- the default-physics one where the flags are data, the
    # nitrogen-cycle one, and their optional-physics instantiations (run-time flags) for the rest
    family = ("stepCoopNKernel<" if name == "nitrogen" else "stepCoopNXKernel<" if name == "everything" else
              "stepCoopKernel<" if name in ("no_water_hresp", "soil_phenol", "calendar_phenology", "no_events") else "stepCoopXKernel<")
    assert kernel.startswith(family), (name, kernel)
    compare(f"{name} [{kernel}]", got, state, want, final)
    # ... and the one-wave kernel's run-time-flag build, which bigger batches and full-state launches of these sets take
    got, state, status, kernel = lean_run(flags, clim60, members70, ev, kernel=sa.KERNEL_ONE_WAVE)
    assert kernel.startswith("stepFastKernel<double"), kernel
    compare(f"{name} [{kernel}]", got, state, want, final)


PHENOLOGY_MODES = {"gdd": dict(), "soil_phenol": dict(gdd=0, soilPhenol=1), "calendar": dict(gdd=0),
                   "russell_4": dict(events=0, gdd=0, soilPhenol=1), "no_water_hresp": dict(waterHResp=0)}


@pytest.mark.parametrize("mode", list(PHENOLOGY_MODES))
def test_phenology_mode_and_events_are_data_for_the_compiled_in_flag_sets(mode, oracle, clim60, members70):
    """events / gdd / soil_phenol / water_hresp do not change the code of the throughput kernels: the plan puts
    the leaf-on variable the flags ask for into the record (and marks every step "no moisture effect" with
    water_hresp off), the parameter conversion the matching threshold into the row the kernels read -- so
    these flag sets (russell_4's among them) take the cooperative kernels, every layout of them, the one-wave
    kernel's default-flag build and the nitrogen-cycle kernel, against the oracle"""
    def run(flags, kernel, ev):
        b = sa.Batch(flags, 1, members70.shape[0], sa.F64, fast_math=True, kernel=kernel)
        if ev is not None:
            b.set_events(0, ev)
        b.set_climate(0, clim60)
        b.set_params(0, members70)
        b.setup()
        planes, _ = b.run()
        out = planes.cpu().numpy(), b.get_state(), b.last_launch()["kernel"]
        assert (np.asarray(b.get_status()) == 0).all()
        b.close()
        return out
    flags = _flags(**PHENOLOGY_MODES[mode])
    ev = _events_all_types(clim60) if flags[F_EVENTS] else None
    want, final, st = oracle.run_block(flags, members70, clim60, ev)
    assert (st == 0).all()
    got = {}
    # (members70's dVpdExp is not 2: the general-exponent builds)
    for kernel, expect in ((sa.KERNEL_AUTO, "stepCoopKernel<double, false, true, false>"),
                           (sa.KERNEL_COOP_PAIR, "stepCoopPairKernel<double, false, false>"),
                           (sa.KERNEL_COOP_QUAD, "stepCoopQuadKernel<double, false>"),
                           (sa.KERNEL_ONE_WAVE, "stepFastKernel<double, false, 0, 1, false>")):
        planes, state, name = run(flags, kernel, ev)
        assert name == expect, name
        compare(f"{mode} {name}", planes, state, want, final)
        got[kernel] = planes
    np.testing.assert_array_equal(got[sa.KERNEL_AUTO], got[sa.KERNEL_COOP_PAIR])
    np.testing.assert_array_equal(got[sa.KERNEL_AUTO], got[sa.KERNEL_COOP_QUAD])
    if mode == "no_water_hresp":
        return                     # (anaerobic needs water_hresp: context.c:203-212)
    nflags = _flags(litterPool=1, anaerobic=1, nitrogenCycle=1, **PHENOLOGY_MODES[mode])
    want, final, st = oracle.run_block(nflags, members70, clim60, ev)
    planes, state, name = run(nflags, sa.KERNEL_AUTO, ev)
    assert name.startswith("stepCoopNKernel<double"), name
    compare(f"{mode} {name}", planes, state, want, final)


def test_generic_throughput_kernel_agrees_with_strict_kernel(clim60, members70):
    flags = _flags(**FLAG_SETS["everything"])
    ev = _events_all_types(clim60)
    g_fast, s_fast, _, _ = lean_run(flags, clim60, members70, ev, fast=True)
    g_strict, s_strict, _, _ = lean_run(flags, clim60, members70, ev, fast=False)
    d = np.abs(g_fast - g_strict).max()
    print("generic throughput vs strict kernel: max|d|", d)
    assert d < 1e-9
    assert np.allclose(s_fast[:, :13], s_strict[:, :13], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", ["nitrogen", "everything", "soil_phenol", "carbon_saturation"])
def test_fp32_mixed_with_optional_flags(name, oracle, clim60, members70):
    """fp32 flux arithmetic over fp64 pools: same tolerance class as the default-flag fp32 test."""
    flags = _flags(**FLAG_SETS[name])
    ev = _events_all_types(clim60)
    got, state, status, _ = lean_run(flags, clim60, members70, ev, prec=sa.F32_MIXED)
    want, final, st = oracle.run_block(flags, members70, clim60, ev)
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    rel = (np.abs(got - want) / scale).max()
    print(f"fp32-mixed, {name}: max |d| / max|plane|", rel)
    assert np.isfinite(got).all()
    assert rel < 2e-3


# ---- the cooperative kernel of the nitrogen-cycle flag set (stepCoopNKernel) ----------------------
NCYCLE = dict(litterPool=1, anaerobic=1, nitrogenCycle=1)


def _ncycle_scenario(clim60, members70, lethal=True):
    """the year of the flag-set tests with every event type, plus what the cooperative kernel's rare
    hand-overs need: a clear-cut that kills every stand and a re-planting (death block), a member dead
    from the start, members starved of nitrogen (tiny mineral pool, no fixation, small storage: the
    exact-supply round trip of checkNitrogenLimitation), a ragged second chunk"""
    ev = _events_all_types(clim60)
    if lethal:
        def add(day, typ, *p):
            e = sa.Event(); e.type = typ; e.year = int(clim60.year[0]); e.day = day
            for i, v in enumerate(p):
                e.p[i] = v
            ev.append(e)
        add(230, 1, 1.0, 1.0, 0.0, 0.0)          # clear-cut: every member dies
        add(240, 3, 40.0, 300.0, 50.0, 60.0)     # re-planting
        ev.sort(key=lambda e: (e.year, e.day))
    members = members70.copy()
    if lethal:
        members[5, param_index("plantWoodInit")] = 0.0
    for m in (2, 9, 40, 66):                      # productive stands whose uptake demand exceeds the mineral pool
        members[m] = members70[0]                 # (member 0 = the unperturbed parameter file)
        members[m, param_index("aMax")] *= 3.0    # (the base stand loses carbon: no creation, no nitrogen demand)
        members[m, param_index("baseVegResp")] *= 0.3
        if m != 2:
            members[m, param_index("mineralNInit")] = 1e-4 * (1 + m)
            members[m, param_index("plantStorageNInit")] = 0.02
            members[m, param_index("soilOrgNInit")] *= 0.01
            members[m, param_index("litterOrgNInit")] *= 0.01
    return ev, members


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_nitrogen_cycle_cooperative_kernel_against_the_oracle(prec, oracle, clim60, members70):
    """stepCoopNKernel forced, launch cut at odd steps (a one-step piece and a tile tail included):
    fp64 at 1e-9 on planes and final pools with the lethal events, fp32-mixed at 2e-6 of the plane
    maximum without them; the nitrogen-starved members do get limited (their wood creation differs from
    an unstarved twin's), i.e. the slow hand-over ran"""
    flags = _flags(**NCYCLE)
    ev, members = _ncycle_scenario(clim60, members70, lethal=prec == sa.F64)
    T = clim60.n_steps
    b = sa.Batch(flags, 1, members.shape[0], prec, fast_math=True if prec == sa.F64 else None, kernel=sa.KERNEL_COOP_NCYCLE)
    b.set_events(0, ev)
    b.set_climate(0, clim60)
    b.set_params(0, members)
    b.setup()
    planes, _ = b.alloc_outputs(T)
    for a, z in ((0, 1), (1, 7), (7, 6000), (6000, 6015), (6015, T)):
        b.run(a, z - a, planes=planes[:, a:z])
    li = b.last_launch()
    got = planes.double().cpu().numpy()
    state, status = b.get_state(), b.get_status()
    b.close()
    assert li["kernel"] == "stepCoopNKernel<%s, false>" % ("double" if prec == sa.F64 else "float"), li
    want, final, st = oracle.run_block(flags, members, clim60, ev)
    assert (st == 0).all() and (np.asarray(status) == 0).all()
    if prec == sa.F64:
        compare("ncycle coop f64", got, state, want, final)
        assert state[0, 30] >= 0 and state[0, 0] > 100.0          # died at the clear-cut, wood is back
    else:
        scale = np.abs(want).max(axis=(1, 2), keepdims=True)
        assert (np.abs(got - want) / scale).max() < 2e-6
    # the productive members WERE limited (the exact-supply hand-over between the carbon and the soil wave ran):
    # with nitrogen in abundance the oracle grows more wood on some steps
    for m in (2, 40):
        rich = members[m].copy()
        rich[param_index("mineralNInit")] = 1e4
        rich[param_index("plantStorageNInit")] = 1e4
        _, rec_poor, _ = oracle.run_member(flags, members[m], clim60, ev)
        _, rec_rich, _ = oracle.run_member(flags, rich, clim60, ev)
        assert (rec_rich[:, 11] - rec_poor[:, 11] > 1e-7).sum() > 100          # woodCreation, column 11


def test_nitrogen_cycle_kernel_paths_give_the_same_bits(clim60, members70):
    """regular 16-step tiles against the general step (SIPNET_KOPT_NO_REGULAR_TILES), and the cooperative
    kernel against a re-run of itself: identical planes, state and rings (which path a wavefront takes
    depends on its neighbours); against the one-wave kernel of the same flag set: to rounding"""
    flags = _flags(**NCYCLE)
    ev, members = _ncycle_scenario(clim60, members70, lethal=False)
    members = np.concatenate([members, members[:58]])                 # 128 members: two full chunks
    members[70, param_index("plantWoodInit")] = 0.0                   # chunk 1 never leaves the general step
    T = clim60.n_steps
    outs = {}
    for key, kernel, opt in (("reg", sa.KERNEL_COOP_NCYCLE, 0), ("gen", sa.KERNEL_COOP_NCYCLE, sa.KOPT_NO_REGULAR_TILES),
                             ("again", sa.KERNEL_COOP_NCYCLE, 0), ("pair", sa.KERNEL_COOP_NCYCLE_PAIR, 0),
                             ("pair_gen", sa.KERNEL_COOP_NCYCLE_PAIR, sa.KOPT_NO_REGULAR_TILES), ("one", sa.KERNEL_ONE_WAVE, 0)):
        b = sa.Batch(flags, 1, members.shape[0], sa.F64, fast_math=True, kernel=kernel, kernel_options=opt)
        b.set_climate(0, clim60)
        b.set_params(0, members)
        b.setup()
        planes, _ = b.alloc_outputs(T)
        for a, z in ((0, 5), (5, 8003), (8003, T)):
            b.run(a, z - a, planes=planes[:, a:z])
        outs[key] = (planes.cpu().numpy(), b.get_state(), b.get_rings())
        b.close()
    for k in range(3):
        np.testing.assert_array_equal(outs["reg"][k], outs["gen"][k])
        np.testing.assert_array_equal(outs["reg"][k], outs["again"][k])
        np.testing.assert_array_equal(outs["reg"][k], outs["pair"][k])         # two chunks per eight-wave workgroup
        np.testing.assert_array_equal(outs["reg"][k], outs["pair_gen"][k])
    assert np.array_equal(outs["reg"][0][:, :, 3], outs["reg"][0][:, :, 73])      # the same member in both chunks
    assert np.abs(outs["reg"][0] - outs["one"][0]).max() < 1e-11


# ---- the optional-physics instantiations of the cooperative layouts (stepCoopXKernel & co: run-time flags) ----------
OPT_SETS = {
    "russell_3": (dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0), "stepCoopX"),
    "anaerobic_litter_saturation_flooding": (dict(anaerobic=1, litterPool=1, carbonSaturation=1, flooding=1), "stepCoopX"),
    "growth_resp_leaf_water": (dict(growthResp=1, leafWater=1), "stepCoopX"),
    "anaerobic_alone": (dict(anaerobic=1), "stepCoopX"),
    "everything": (FLAG_SETS["everything"], "stepCoopNX"),
}


@pytest.mark.parametrize("name", list(OPT_SETS))
def test_optional_physics_layouts_give_the_same_bits_and_match_the_oracle(name, oracle, clim60, members70):
    """every cooperative layout of an optional flag set -- ring in LDS / in HBM, two chunks per workgroup, regular
    tiles against the general step -- forced in turn on a year with every event type, a clear-cut that kills every
    stand, a re-planting, a member dead from the start and a launch cut at odd steps: identical planes, state and
    rings (which path a wavefront takes depends on its neighbours), 1e-9 against the oracle, the one-wave kernel's
    run-time-flag build to rounding"""
    kw, family = OPT_SETS[name]
    flags = _flags(**kw)
    ncyc = family == "stepCoopNX"
    ev, members = _ncycle_scenario(clim60, members70, lethal=True) if ncyc else (None, None)
    if not ncyc:
        ev = _events_all_types(clim60)
        def add(day, typ, *p):
            e = sa.Event(); e.type = typ; e.year = int(clim60.year[0]); e.day = day
            for i, v in enumerate(p):
                e.p[i] = v
            ev.append(e)
        add(230, 1, 1.0, 1.0, 0.0, 0.0)          # clear-cut: every member dies
        add(240, 3, 40.0, 300.0, 50.0, 60.0)     # re-planting
        ev.sort(key=lambda e: (e.year, e.day))
        members = members70.copy()
        members[5, param_index("plantWoodInit")] = 0.0
    members = np.concatenate([members, members[:58]])                 # 128 members: two full chunks
    T = clim60.n_steps
    one, two = (sa.KERNEL_COOP_NCYCLE, sa.KERNEL_COOP_NCYCLE_PAIR) if ncyc else (sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_PAIR)
    runs = [("reg", one, 0), ("gen", one, sa.KOPT_NO_REGULAR_TILES), ("pair", two, 0), ("pair_gen", two, sa.KOPT_NO_REGULAR_TILES),
            ("one_wave", sa.KERNEL_ONE_WAVE, 0)]
    if not ncyc:
        runs.insert(2, ("hbm", sa.KERNEL_COOP_HBM, 0))
    outs, names = {}, {}
    for key, kernel, opt in runs:
        b = sa.Batch(flags, 1, members.shape[0], sa.F64, fast_math=True, kernel=kernel, kernel_options=opt)
        b.set_events(0, ev)
        b.set_climate(0, clim60)
        b.set_params(0, members)
        b.setup()
        planes, _ = b.alloc_outputs(T)
        for a, z in ((0, 1), (1, 7), (7, 8003), (8003, 8019), (8019, T)):
            b.run(a, z - a, planes=planes[:, a:z])
        names[key] = b.last_launch()["kernel"]
        assert (np.asarray(b.get_status()) == 0).all()
        outs[key] = (planes.cpu().numpy(), b.get_state(), b.get_rings())
        b.close()
    assert names["reg"].startswith(family + "Kernel<double"), names
    assert names["pair"].startswith(family + "PairKernel<double"), names
    assert names["one_wave"].startswith("stepFastKernel<double, false, 1,"), names
    want, final, st = oracle.run_block(flags, members, clim60, ev)
    assert (st == 0).all()
    compare(f"{name} {names['reg']}", outs["reg"][0], outs["reg"][1], want, final)
    for key in outs:
        if key in ("reg", "one_wave"):
            continue
        for k in range(3):
            np.testing.assert_array_equal(outs["reg"][k], outs[key][k], err_msg=f"{key} [{k}]")
    assert np.abs(outs["reg"][0] - outs["one_wave"][0]).max() < 1e-10
    assert outs["reg"][1][0, 30] >= 0 and outs["reg"][1][0, 0] > 100.0          # died at the clear-cut, wood is back


@pytest.mark.parametrize("name", ["russell_3", "anaerobic_litter_saturation_flooding"])
def test_optional_physics_layouts_in_fp32_mixed(name, oracle, clim60, members70):
    kw, family = OPT_SETS[name]
    flags = _flags(**kw)
    ev = _events_all_types(clim60)
    want, final, st = oracle.run_block(flags, members70, clim60, ev)
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    got = {}
    for kernel in (sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_PAIR):
        b = sa.Batch(flags, 1, members70.shape[0], sa.F32_MIXED, kernel=kernel)
        b.set_events(0, ev)
        b.set_climate(0, clim60)
        b.set_params(0, members70)
        b.setup()
        planes, _ = b.run()
        assert b.last_launch()["kernel"].startswith(family), b.last_launch()
        got[kernel] = planes.double().cpu().numpy()
        b.close()
        rel = (np.abs(got[kernel] - want) / scale).max()
        print(f"fp32-mixed {name} kernel {kernel}: max |d| / max|plane| = {rel:.3e}")
        assert rel < 2e-3
    np.testing.assert_array_equal(got[sa.KERNEL_COOP_LDS], got[sa.KERNEL_COOP_PAIR])
