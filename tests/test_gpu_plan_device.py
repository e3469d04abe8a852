"""The site plan built on the device (csrc/plan_device.h) against the host builder (csrc/plan.cpp).

For a site without events / resumed checkpoint whose steps are all >= 0.0202 days the 256-byte per-step records and
the ring's eviction list are produced by four kernels from the uploaded climate.  These tests download them and compare
them BYTE BY BYTE with buildSitePlan()'s (sipnet_debug_plan_compare), over forcings that exercise every branch of
the sequential parts: constant step lengths (one run descriptor), the reference's own half-daily and half-hourly
files, random lengths, runs of random lengths, ring resets (steps of 5 days and more), several years, a year that steps
back, sites of different lengths (partial last tiles), every phenology mode, narrow records of fp32-mixed batches.
They also check what is left to the host (events, short steps), the run-time fill of log2(vpd), and that the step
kernels' results do not depend on who built the plan.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd._lib import Event, lib
from sipnet_amd.config import param_index as pi
from sipnet_amd.io import ClimTable
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


def year_clim(n=17520, site=0):
    return synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(n, site=site)))


def with_lengths(clim, lengths, years=None):
    d = clim.data[:len(lengths)].copy()
    d[:, 0] = lengths
    y = clim.year[:len(lengths)].copy() if years is None else np.asarray(years, dtype=np.int32)
    return ClimTable(d, y, clim.day[:len(lengths)].copy())


def compare(b, site, ignore_log2=True):
    nr, no = C.c_int64(), C.c_int64()
    fs, fo = C.c_int32(), C.c_int32()
    info = (C.c_int32 * 8)()
    sa._lib.check(lib().sipnet_debug_plan_compare(b.h, site, int(ignore_log2), C.byref(nr), C.byref(no), C.byref(fs),
                                                  C.byref(fo), info), "plan_compare")
    return dict(records=nr.value, ops=no.value, first_step=fs.value, first_offset=fo.value, runs=info[0], n_ops=info[1],
                status=info[2], status_at=info[3], ring_walk_us=info[4] / 100.0, room=info[5])


def forcings():
    rng = np.random.default_rng(20260503)
    yc = year_clim()
    out = {"half-hourly year": yc, "half-hourly, partial last tile": yc.slice(0, 17520 - 7), "one tile": yc.slice(0, 16), "three records": yc.slice(0, 3)}
    for case in ("niwot", "russell_1"):
        out[case] = helpers.load_smoke_case(case)["clim"]
    n = 6000
    out["random lengths"] = with_lengths(yc, rng.choice([1 / 48, 1 / 24, 0.3, 0.5, 1.0, 2.0, 3.0, 6.0], n))
    out["uniform random lengths"] = with_lengths(yc, rng.uniform(0.0202, 1.5, n))
    runs = np.concatenate([np.full(int(rng.integers(1, 400)), float(rng.choice([1 / 48, 1 / 24, 0.25, 1.0, 2.5, 0.1, 5.0, 7.5]))) for _ in range(60)])
    out["runs of lengths, resets"] = with_lengths(yc, runs[:17000])
    out["long steps"] = with_lengths(yc, np.concatenate([np.full(200, 3.0), np.full(200, 2.0), np.full(100, 4.9), np.full(300, 0.05), np.full(5, 4.0)]))
    years = 2001 + np.arange(17520) // 3000
    years[9000:9100] -= 2                      # a forcing whose years step back (summariseTile's maxima, the prefix maximum)
    out["several years, one stepping back"] = with_lengths(yc, np.full(17520, 1 / 48), years)
    return out


FORCINGS = forcings() if torch.cuda.is_available() else {}


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32mixed"])
@pytest.mark.parametrize("flagset", ["default", "soil_phenol", "day_of_year", "no_water_hresp"])
def test_device_built_records_are_the_host_builders_bytes(base, prec, flagset):
    flags = {"default": sa.flags_from(), "soil_phenol": sa.flags_from(gdd=0, soilPhenol=1), "day_of_year": sa.flags_from(gdd=0),
             "no_water_hresp": sa.flags_from(waterHResp=0)}[flagset]
    names = list(FORCINGS)
    # (KOPT_DEVICE_PLAN: the irregular forcings too -- by default their ring schedules are left to the host's cores)
    b = sa.Batch(flags, len(names), 64, prec, fast_math=True if prec == sa.F64 else None, kernel_options=sa.KOPT_DEVICE_PLAN)
    b.set_climates([FORCINGS[k] for k in names])
    b.set_params(None, base)
    b.setup()
    assert b.last_launch()["plan_device_sites"] == len(names)
    for s, k in enumerate(names):
        r = compare(b, s)
        assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, (k, r)
    r = compare(b, names.index("half-hourly year"))
    assert r["runs"] == 1, r                # ~245 walked steps and one descriptor
    assert compare(b, names.index("niwot"))["runs"] == 0
    b.close()


def test_log2_vpd_is_filled_from_the_hosts_values_when_a_member_reads_it(base):
    clims = [FORCINGS["half-hourly year"], FORCINGS["niwot"]]
    odd = np.tile(base, (64, 1))
    odd[5, pi("dVpdExp")] = 2.5
    # (a) such a member is known when the plan is built, (b) it appears afterwards: filled before the next launch
    for late in (False, True):
        b = sa.Batch(sa.flags_from(), 2, 64, sa.F64, fast_math=True, kernel_options=sa.KOPT_DEVICE_PLAN)
        for s, c in enumerate(clims):
            b.set_climate(s, c)
        b.set_params(None, base if late else odd)
        b.setup()
        if late:
            assert compare(b, 0, ignore_log2=False)["records"] > 0      # not filled: nobody reads it
            b.set_params(None, odd)
            b.run(0, 4)
        for s in range(2):
            r = compare(b, s, ignore_log2=False)
            assert r["records"] == 0 and r["ops"] == 0, (late, s, r)
        b.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32mixed"])
def test_results_do_not_depend_on_who_built_the_plan(base, prec):
    clims = [FORCINGS["half-hourly year"].slice(0, 3000), FORCINGS["niwot"], FORCINGS["runs of lengths, resets"].slice(0, 2500)]
    members = synth.perturbed_params(base, 128, seed=77)
    members[3, pi("dVpdExp")] = 1.7
    planes = {}
    for opt in (sa.KOPT_DEVICE_PLAN, sa.KOPT_HOST_PLAN):
        b = sa.Batch(sa.flags_from(), len(clims), 128, prec, fast_math=True if prec == sa.F64 else None, kernel_options=opt)
        if opt == sa.KOPT_HOST_PLAN:
            for s, c in enumerate(clims):
                b.set_climate(s, c)
        else:
            b.set_climates(clims[1:], first_site=1)      # (the bulk hand-over, and a single one)
            b.set_climate(0, clims[0])
        for s in range(len(clims)):
            b.set_params(s, members)
        b.setup()
        assert b.last_launch()["plan_device_sites"] == (0 if opt == sa.KOPT_HOST_PLAN else len(clims))
        out, _ = b.run()
        # (rows past a shorter site's last record are not written)
        planes[opt] = [out[:, :c.n_steps, s * 128:(s + 1) * 128].clone() for s, c in enumerate(clims)]
        series = [b.site_series(s) for s in range(len(clims))]
        planes[(opt, "series")] = series
        planes[(opt, "state")] = b.get_state().copy()
        b.close()
    for x, y in zip(planes[sa.KOPT_DEVICE_PLAN], planes[sa.KOPT_HOST_PLAN]):
        assert torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8))   # (bit patterns: NaNs included)
    assert np.array_equal(planes[(sa.KOPT_DEVICE_PLAN, "state")], planes[(sa.KOPT_HOST_PLAN, "state")], equal_nan=True)
    for (g0, d0), (g1, d1) in zip(planes[(sa.KOPT_DEVICE_PLAN, "series")], planes[(sa.KOPT_HOST_PLAN, "series")]):
        assert np.array_equal(g0, g1) and np.array_equal(d0, d1)


def test_sites_the_host_keeps(base):
    yc = FORCINGS["half-hourly year"].slice(0, 2000)
    short = with_lengths(yc, np.full(2000, 0.0201))                # below the bound that rules a ring overflow out: the host's
    ev = Event(type=2, year=int(yc.year[10]), day=int(yc.day[10]), pad=0, p=(C.c_double * 4)(1.0, 0.0, 0.0, 0.0))   # SIPNET_EV_IRRIG
    for flags, want in ((sa.flags_from(events=1), 2), (sa.flags_from(events=0), 2)):   # (events: the device's too; switched off: the list is ignored)
        b = sa.Batch(flags, 3, 64, sa.F64, fast_math=True)
        b.set_climate(0, yc)
        b.set_climate(1, short)
        b.set_climate(2, yc)
        b.set_events(2, [ev])
        b.set_params(None, base)
        b.setup()
        assert b.last_launch()["plan_device_sites"] == want
        assert compare(b, 0)["records"] == 0
        b.run(0, 100)
        b.close()


def test_the_default_leaves_forcings_without_long_runs_of_equal_steps_to_the_host(base):
    """one lane walks the ring's schedule outside runs of equal step lengths (~0.5 us a step): niwot's half-daily records
    (no two neighbours alike) are the host's, a half-hourly year (245 walked steps) the device's"""
    names = ["half-hourly year", "niwot", "russell_1", "uniform random lengths", "several years, one stepping back"]
    b = sa.Batch(sa.flags_from(), len(names), 64, sa.F64, fast_math=True)
    b.set_climates([FORCINGS[k] for k in names])
    b.set_params(None, base)
    b.setup()
    assert b.last_launch()["plan_device_sites"] == 3
    for s in (0, 2, 4):
        assert compare(b, s)["records"] == 0
    for s in (1, 3):
        with pytest.raises(Exception):
            compare(b, s)
    b.run(0, 64)
    b.close()


def test_forcings_back_to_back_on_one_batch(base):
    """the second forcing's climate copy and plan kernels behind the first's launches; buffers reused and regrown"""
    b = sa.Batch(sa.flags_from(), 2, 64, sa.F64, fast_math=True, kernel_options=sa.KOPT_DEVICE_PLAN)
    b.set_params(None, base)
    seq = [(FORCINGS["half-hourly year"].slice(0, 4000), FORCINGS["niwot"]),
           (FORCINGS["random lengths"], FORCINGS["half-hourly year"].slice(100, 3000)),
           (FORCINGS["half-hourly year"], FORCINGS["runs of lengths, resets"])]
    for a, c in seq:
        b.set_climate(0, a)
        b.set_climate(1, c)
        b.setup()
        b.run(0, min(a.n_steps, c.n_steps, 500))
        for s in range(2):
            r = compare(b, s)
            assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, r
    b.close()


def tillage_events(clim, steps):
    """a tillage and an irrigation event on the records `steps` of the forcing (SIPNET_EV_TILL = 4, SIPNET_EV_IRRIG = 2)"""
    out = []
    for k, t in enumerate(steps):
        typ, p = (4, (0.5, 0.0, 0.0, 0.0)) if k % 2 == 0 else (2, (1.5, 1.0, 0.0, 0.0))
        out.append(Event(type=typ, year=int(clim.year[t]), day=int(clim.day[t]), pad=0, p=(C.c_double * 4)(*p)))
    return out


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32mixed"])
def test_sites_with_events_are_built_on_the_device_too(base, prec):
    """the host's pass hands the events per record and the tillage modifier's decay over (24 bytes a step); records, evictions
    and event records are the host builder's bytes, and so are the results"""
    ru = helpers.load_smoke_case("russell_1")            # the reference's own events file on its own forcing
    yc = FORCINGS["half-hourly year"]
    # (one event per day at most lands on distinct records only with one record a day; half-hourly: the day's first record)
    clims = [ru["clim"], yc, yc.slice(0, 9000)]
    events = [ru["events"], tillage_events(yc, [48 * 40, 48 * 41, 48 * 200, 48 * 300]), []]
    members = synth.perturbed_params(ru["params"], 64, seed=5)
    out = {}
    for opt in (sa.KOPT_DEVICE_PLAN, sa.KOPT_HOST_PLAN):
        b = sa.Batch(ru["flags"], 3, 64, prec, fast_math=True if prec == sa.F64 else None, kernel_options=opt)
        for s in range(3):
            b.set_events(s, events[s])
        b.set_climates(clims)
        b.set_params(None, members)
        b.setup()
        if opt == sa.KOPT_DEVICE_PLAN:
            assert b.last_launch()["plan_device_sites"] == 3
            for s in range(3):
                r = compare(b, s)
                assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, (s, r)
        pl, _ = b.run()
        out[opt] = ([pl[:, :c.n_steps, s * 64:(s + 1) * 64].clone() for s, c in enumerate(clims)], b.get_state().copy(),
                    [b.site_series(s) for s in range(3)])
        b.close()
    for x, y in zip(out[sa.KOPT_DEVICE_PLAN][0], out[sa.KOPT_HOST_PLAN][0]):
        assert torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8))
    assert np.array_equal(out[sa.KOPT_DEVICE_PLAN][1], out[sa.KOPT_HOST_PLAN][1], equal_nan=True)
    for (g0, d0), (g1, d1) in zip(out[sa.KOPT_DEVICE_PLAN][2], out[sa.KOPT_HOST_PLAN][2]):
        assert np.array_equal(g0, g1) and np.array_equal(d0, d1)
    assert any(d.max() > 0 for _, d in out[sa.KOPT_HOST_PLAN][2])          # a tillage modifier was in effect


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32mixed"])
def test_a_resumed_segment_is_built_on_the_device_too(base, prec):
    """the ring a checkpoint hands over becomes the walk's initial queue (its entries: insert step 0, their own weights); GDD,
    year roll-overs and the tillage modifier continue from the checkpoint's"""
    yc = FORCINGS["half-hourly year"]
    irregular = FORCINGS["runs of lengths, resets"]
    cases = [(yc, 5000, []), (yc, 100, tillage_events(yc, [48 * 1])), (irregular, 3333, []), (FORCINGS["niwot"], 1000, [])]
    members = synth.perturbed_params(base, 64, seed=9)
    for clim, k, ev in cases:
        b1 = sa.Batch(sa.flags_from(), 1, 64, prec, fast_math=True if prec == sa.F64 else None, kernel_options=sa.KOPT_DEVICE_PLAN)
        b1.set_events(0, ev)
        b1.set_climate(0, clim.slice(0, k))
        b1.set_params(0, members)
        b1.setup()
        b1.run(0, k)
        cks = [b1.export_restart(0, m, k) for m in range(64)]
        b1.close()
        seg2 = clim.slice(k, min(clim.n_steps, k + 6000))
        res = {}
        for opt in (sa.KOPT_DEVICE_PLAN, sa.KOPT_HOST_PLAN):
            b2 = sa.Batch(sa.flags_from(), 1, 64, prec, fast_math=True if prec == sa.F64 else None, kernel_options=opt)
            b2.set_climate(0, seg2)
            b2.set_params(0, members)
            b2.set_resume(0, cks[0])
            b2.setup()
            if opt == sa.KOPT_DEVICE_PLAN:
                assert b2.last_launch()["plan_device_sites"] == 1
                r = compare(b2, 0)
                assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, (k, r)
            b2.import_restart(0, cks)
            pl, _ = b2.run()
            res[opt] = (pl.clone(), b2.get_state().copy())
            b2.close()
        assert torch.equal(res[sa.KOPT_DEVICE_PLAN][0].view(torch.uint8), res[sa.KOPT_HOST_PLAN][0].view(torch.uint8))
        assert np.array_equal(res[sa.KOPT_DEVICE_PLAN][1], res[sa.KOPT_HOST_PLAN][1], equal_nan=True)


def test_a_resumed_ring_of_many_entries_fits_its_eviction_list(base):
    """a checkpoint's ring of ~241 half-hourly entries followed by a segment with a step long enough to evict them all at
    once: 2 n + preK evictions in the worst case, more than a fresh ring's 2 n -- the list's room follows the checkpoint
    (engine.hip devRingOpRoom), the walk never writes past it, and the NEXT site's list is the host builder's bytes too"""
    yc = FORCINGS["half-hourly year"]
    members = synth.perturbed_params(base, 64, seed=11)
    b1 = sa.Batch(sa.flags_from(), 1, 64, sa.F64, fast_math=True)
    b1.set_climate(0, yc.slice(0, 400))
    b1.set_params(0, members)
    b1.setup()
    b1.run(0, 400)
    ck = b1.export_restart(0, 0, 400)
    b1.close()
    half = 1 / 48
    tails = {"n = 1, one 4.9-day step": [4.9], "a daily step": [1.0, half, half], "999 half-hourly + one 4-day step": [half] * 999 + [4.0],
             "alternating": [half, 4.0, half, 4.5] * 20}
    names = list(tails)
    clims = [with_lengths(yc.slice(400, 400 + len(tails[k])), tails[k]) for k in names]
    b2 = sa.Batch(sa.flags_from(), len(names), 64, sa.F64, fast_math=True, kernel_options=sa.KOPT_DEVICE_PLAN)
    b2.set_climates(clims)
    b2.set_params(None, members)
    for s in range(len(names)):
        b2.set_resume(s, ck)
    b2.setup()
    assert b2.last_launch()["plan_device_sites"] == len(names)
    for s, k in enumerate(names):
        r = compare(b2, s)
        n = len(tails[k])
        assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0 and r["n_ops"] <= r["room"], (k, r)
        assert r["room"] > 2 * n + 8 + 200, (k, r)       # (the checkpoint's entries are in the bound)
    assert compare(b2, 0)["n_ops"] > 2 * 1 + 8           # n = 1: the old room (10) would not have held the list
    b2.close()


def test_a_resumed_ring_that_does_not_carry_the_window_is_the_hosts(base):
    """the device walk assumes the live weights sum to the 5-day window (it cannot run empty or fill up then); a checkpoint
    whose ring does not is left to the host builder, which treats it as the reference does"""
    yc = FORCINGS["half-hourly year"]
    b1 = sa.Batch(sa.flags_from(), 1, 64, sa.F64, fast_math=True)
    b1.set_climate(0, yc.slice(0, 400))
    b1.set_params(0, base)
    b1.setup()
    b1.run(0, 400)
    ck = b1.export_restart(0, 0, 400)
    b1.close()
    ck.mean_weights[ck.mean_last] += 0.004           # a little more than the window
    b2 = sa.Batch(sa.flags_from(), 1, 64, sa.F64, fast_math=True, kernel_options=sa.KOPT_DEVICE_PLAN)
    b2.set_climate(0, yc.slice(400, 2000))
    b2.set_params(0, base)
    b2.set_resume(0, ck)
    b2.setup()
    assert b2.last_launch()["plan_device_sites"] == 0
    b2.close()


def test_random_forcings_against_the_host_builder(base):
    """60 random forcings (lengths in runs of random length, resets, years rolling over or stepping back at random records,
    random tillage / irrigation events, random phenology flags, both precisions), eight sites a batch: records, evictions and
    event records byte for byte"""
    rng = np.random.default_rng(2026100305)
    yc = FORCINGS["half-hourly year"]
    pool = [1 / 48, 1 / 24, 0.0202, 0.125, 0.25, 0.5, 1.0, 2.0, 3.5, 4.9999, 5.0, 9.0]
    total = with_events = 0
    for trial in range(8):
        clims, events = [], []
        for s in range(8):
            n = int(rng.integers(1, 3000))
            lens = np.concatenate([np.full(int(rng.integers(1, 600)), pool[int(rng.integers(len(pool)))] if rng.random() < 0.8 else float(rng.uniform(0.0202, 6.0)))
                                   for _ in range(40)])[:n]
            n = len(lens)
            years = 2000 + np.cumsum(rng.random(n) < 0.002).astype(np.int32)
            if rng.random() < 0.3 and n > 10:
                k = int(rng.integers(1, n))
                years[k:k + int(rng.integers(1, 50))] -= int(rng.integers(1, 3))
            c = with_lengths(yc, lens, years)
            clims.append(c)
            ev = []
            if rng.random() < 0.5 and n > 4:
                days = sorted(set(int(x) for x in rng.integers(0, n, size=int(rng.integers(1, 6)))))
                seen = set()
                for t in days:                      # (events must ascend in time and match a record: the first record of a day)
                    key = (int(c.year[t]), int(c.day[t]))
                    t0 = int(np.argmax((c.year == key[0]) & (c.day == key[1])))
                    if key in seen or (ev and (key[0], key[1]) <= (ev[-1].year, ev[-1].day)) or t0 != t and (t0 > 0 and (int(c.year[t0 - 1]), int(c.day[t0 - 1])) >= key):
                        continue
                    seen.add(key)
                    ev += tillage_events(c, [t0]) if rng.random() < 0.6 else tillage_events(c, [t0, t0])[1:]
            events.append(ev)
        flags = [sa.flags_from(), sa.flags_from(gdd=0, soilPhenol=1), sa.flags_from(gdd=0), sa.flags_from(waterHResp=0)][trial % 4]
        prec = sa.F64 if trial % 2 == 0 else sa.F32_MIXED
        b = sa.Batch(flags, 8, 64, prec, fast_math=True if prec == sa.F64 else None, kernel_options=sa.KOPT_DEVICE_PLAN)
        for s in range(8):
            b.set_events(s, events[s])
        b.set_climates(clims)
        b.set_params(None, base)
        try:
            b.setup()
        except Exception:
            # a site-fatal plan (an event the random forcing has no record for): the whole batch is refused, as with the host builder
            b.close()
            continue
        n_dev = b.last_launch()["plan_device_sites"]
        assert n_dev >= 1
        checked = 0
        for s in range(8):
            try:
                r = compare(b, s)
            except Exception:
                continue                      # (a site the prepass left to the host)
            assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, (trial, s, r)
            checked += 1
            with_events += bool(events[s])
        assert checked == n_dev
        total += checked
        b.close()
    print(f"{total} random sites compared, {with_events} of them with events")
    assert total >= 32 and with_events >= 8, (total, with_events)


def test_ten_years_of_half_hourly_records_and_a_single_record(base):
    """175 200 records (one run descriptor of 174 955 steps, ten GDD chains) beside a forcing of ONE record"""
    raw = synth.half_hourly_year_raw(17520)
    one = synth.convert_raw(synth.round_like_file(raw))
    data = np.tile(one.data, (10, 1))
    year = np.concatenate([one.year + k for k in range(10)]).astype(np.int32)
    day = np.tile(one.day, 10)
    long = ClimTable(data, year, day)
    b = sa.Batch(sa.flags_from(), 2, 64, sa.F32_MIXED, kernel_options=sa.KOPT_DEVICE_PLAN)
    b.set_climates([long, one.slice(0, 1)])
    b.set_params(None, base)
    b.setup()
    assert b.last_launch()["plan_device_sites"] == 2
    r = compare(b, 0)
    assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0 and r["runs"] == 1, r
    r = compare(b, 1)
    assert r["status"] == 0 and r["records"] == 0 and r["ops"] == 0, r
    b.run(175200 - 64, 64)
    b.close()
