"""The "Full" instantiations of the throughput kernels: the 44-column
per-step record, every accumulator of the restart schema and the per-member diagnostics counters
(clamp warnings sipnet.c:1346-1356, mass-balance warnings balance.c:122-169) from the kernels
bench.py times -- each compared with the ORACLE (record columns, counters) and with the
strict-order kernel (event-log columns, accumulators)."""
import os

import numpy as np
import pytest

import sipnet_amd as sa
from sipnet_amd import synth
from tests import helpers
from tests.test_gpu_configs import _scenario, year_clim

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")
KERNELS = [("coop_lds", sa.KERNEL_COOP_LDS, 0, "stepCoopKernel<double, true, true, true>"),
           ("coop_hbm", sa.KERNEL_COOP_HBM, 0, "stepCoopKernel<double, true, false, true>"),
           ("coop_pair", sa.KERNEL_COOP_PAIR, 0, "stepCoopPairKernel<double, true, true>"),
           ("one_wave", sa.KERNEL_ONE_WAVE, 0, "stepFastKernel<double, true, 0, 1, true>"),
           ("runtime_flags", sa.KERNEL_ONE_WAVE, sa.KOPT_RUNTIME_FLAGS, "stepFastKernel<double, true, 1, 1, true>"),
           # the optional-physics instantiations (russell_3's flag family + anaerobic, carbon saturation, flooding)
           ("x_lds", sa.KERNEL_COOP_LDS, 0, "stepCoopXKernel<double, false, true, true>"),
           ("x_hbm", sa.KERNEL_COOP_HBM, 0, "stepCoopXKernel<double, false, false, true>"),
           ("x_pair", sa.KERNEL_COOP_PAIR, 0, "stepCoopXPairKernel<double, false, true>"),
           ("x_one_wave", sa.KERNEL_ONE_WAVE, 0, "stepFastKernel<double, false, 1, 1, true>"),
           # the nitrogen-cycle flag set: record and accumulators from the cooperative kernels (the soil wave writes the
           # heterotrophic / soil / nitrogen columns, the carbon wave the plants', the water wave its own)
           ("n_full", sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNFullKernel<double, false>"),
           ("n_pair_full", sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNPairFullKernel<double, false>"),
           ("n_one_wave", sa.KERNEL_ONE_WAVE, 0, "stepFastKernel<double, false, 2, 1, true>"),
           # "everything" (the nitrogen cycle + every other optional flag) with the record: round 5
           ("nx_full", sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNXFullKernel<double, false>"),
           ("nx_pair_full", sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNXPairFullKernel<double, false>"),
           ("nx_auto", sa.KERNEL_AUTO, 0, "stepCoopNXFullKernel<double, false>")]
EVERYTHING = dict(litterPool=1, anaerobic=1, nitrogenCycle=1, carbonSaturation=1, flooding=1, growthResp=1, leafWater=1)
X_FLAGS = dict(anaerobic=1, litterPool=1, carbonSaturation=1, flooding=1, growthResp=1, leafWater=1)


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


def _batch(flags, clim, members, ev, fast, kernel=sa.KERNEL_AUTO, options=0, prec=sa.F64, diag=False):
    b = sa.Batch(flags, 1, members.shape[0], prec, fast_math=fast if prec == sa.F64 else None,
                 kernel=kernel, kernel_options=options)
    if ev is not None:
        b.set_events(0, ev)
    b.set_climate(0, clim)
    b.set_params(0, members)
    if diag:
        b.enable_diagnostics()
    b.setup()
    return b


@pytest.mark.parametrize("name,kernel,options,expect", KERNELS, ids=[k[0] for k in KERNELS])
def test_full_record_from_the_throughput_kernels(name, kernel, options, expect, oracle, base):
    """all 36 `.out` columns of every step against the oracle, the 8 event-log columns and the
    carried accumulators against the strict-order kernel; the launch is split at odd steps"""
    flags = sa.flags_from()
    if name.startswith(("x_", "n_", "nx_")):
        flags = sa.flags_from(**(X_FLAGS if name.startswith("x_") else EVERYTHING if name.startswith("nx_")
                                 else dict(litterPool=1, anaerobic=1, nitrogenCycle=1)))
        base = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    clim, ev, members = _scenario(base, lethal=True)
    members = members[:70]
    if name.startswith(("x_", "nx_")):
        rng = np.random.default_rng(5)
        from sipnet_amd.config import param_index as pi
        members[:, pi("soilCSaturation")] = members[:, pi("soilInit")] * rng.uniform(0.5, 3.0, 70)
        members[:, pi("waterDrainFrac")] = rng.uniform(0.2, 1.0, 70)
        members[:, pi("leafPoolDepth")] *= rng.uniform(0.02, 1.5, 70)
    T = clim.n_steps
    strict = _batch(flags, clim, members, ev, fast=False)
    _, rec_s = strict.run(full=True, want_planes=False)
    rec_s = rec_s.cpu().numpy()
    state_s = strict.get_state()
    strict.close()
    b = _batch(flags, clim, members, ev, fast=True, kernel=kernel, options=options)
    planes, rec = b.alloc_outputs(T, full=True)
    for a, z in ((0, 7), (7, 1000), (1000, 1015), (1015, T)):
        b.run(a, z - a, planes=planes[:, a:z], rec=rec[a:z])
    li = b.last_launch()
    got = rec.cpu().numpy()
    pl = planes.cpu().numpy()
    state = b.get_state()
    b.close()
    assert li["kernel"] == expect, li
    worst = 0.0
    for m in range(members.shape[0]):
        st, want, _ = oracle.run_member(flags, members[m], clim, ev)
        assert st == 0
        scale = np.maximum(np.abs(want).max(axis=0), 1e-3)            # per column
        err = (np.abs(got[:, :36, m] - want) / scale).max()
        worst = max(worst, err)
        assert err < 1e-9, (m, err, int(np.argmax((np.abs(got[:, :36, m] - want) / scale).max(axis=0))))
    # the planes of the same launch are the record's first three columns
    assert np.array_equal(pl[0], got[:, 0]) and np.array_equal(pl[1], got[:, 1]) and np.array_equal(pl[2], got[:, 2])
    # event log (computed leaf-on / leaf-off amounts, plant death) and accumulators vs the strict kernel
    np.testing.assert_allclose(got[:, 36:], rec_s[:, 36:], rtol=1e-9, atol=1e-12)
    assert (got[:, 43] == rec_s[:, 43]).all() and got[:, 43].sum() > 0      # deaths on the same steps
    np.testing.assert_allclose(state[:, 14:27], state_s[:, 14:27], rtol=1e-9, atol=1e-9)
    print("%s: worst record column error %.2e of the column's maximum" % (name, worst))


def test_lean_launches_leave_the_other_accumulators_alone_and_full_state_advances_them(base):
    """without a record / diagnostics only totNee and totGpp advance on the throughput path
    (documented); SIPNET_KOPT_FULL_STATE makes a checkpoint taken after it complete"""
    flags = sa.flags_from()
    clim = year_clim(n=48 * 30, start_day=150)
    members = synth.perturbed_params(base, 64)
    ref = _batch(flags, clim, members, None, fast=False)
    ref.run(want_planes=False)
    s_ref = ref.get_state()
    ck_ref = ref.export_restart(0, 3, clim.n_steps)
    ref.close()
    lean = _batch(flags, clim, members, None, fast=True)
    lean.run(want_planes=False)
    s_lean = lean.get_state()
    lean.close()
    assert (s_lean[:, [15, 16, 17, 18]] == 0).all() and (s_lean[:, 20:27] == 0).all()
    np.testing.assert_allclose(s_lean[:, [14, 19]], s_ref[:, [14, 19]], rtol=1e-9)
    for kernel in (sa.KERNEL_COOP_LDS, sa.KERNEL_ONE_WAVE):
        full = _batch(flags, clim, members, None, fast=True, kernel=kernel, options=sa.KOPT_FULL_STATE)
        full.run(want_planes=False)
        s_full = full.get_state()
        ck = full.export_restart(0, 3, clim.n_steps)
        full.close()
        np.testing.assert_allclose(s_full[:, 14:27], s_ref[:, 14:27], rtol=1e-9, atol=1e-9)
        for k in ("yearlyGpp", "yearlyLitter", "totRtot", "totNpp", "totNee"):
            assert abs(ck.tracker(k) - ck_ref.tracker(k)) <= 1e-9 * max(1.0, abs(ck_ref.tracker(k))), k


SPECIAL = [("strict", False, sa.KERNEL_AUTO, 0), ("coop_lds", True, sa.KERNEL_COOP_LDS, 0),
           ("coop_hbm", True, sa.KERNEL_COOP_HBM, 0), ("coop_pair", True, sa.KERNEL_COOP_PAIR, 0),
           ("one_wave", True, sa.KERNEL_ONE_WAVE, 0),
           ("runtime_flags", True, sa.KERNEL_ONE_WAVE, sa.KOPT_RUNTIME_FLAGS)]


@pytest.mark.parametrize("name,fast,kernel,options", SPECIAL, ids=[k[0] for k in SPECIAL])
def test_diagnostics_counters_match_the_oracle_on_the_special_members(name, fast, kernel, options, oracle, tmp_path):
    """the 16 members of tests/golden/synth walk the rare branches (member 9 trips the snow clamp
    13 010 times, member 7 once, member 8 dies): per-member n_clamp_warn and n_balance_warn equal
    the oracle's sipo_diag, the largest carbon residual stays at rounding level"""
    helpers.gunzip_to(os.path.join(helpers.GOLDEN, "synth", "halfhourly.clim.gz"), str(tmp_path / "hh.clim"))
    clim = sa.read_clim(str(tmp_path / "hh.clim"))
    members = np.load(os.path.join(helpers.GOLDEN, "synth", "members_raw.npy"))
    flags = sa.flags_from()
    want = [oracle.run_member(flags, members[m], clim, want_rec=False)[2] for m in range(members.shape[0])]
    b = _batch(flags, clim, members, None, fast=fast, kernel=kernel, options=options, diag=True)
    T = clim.n_steps
    b.run(0, T // 3, want_planes=False)                 # counters accumulate over launches
    b.run(T // 3, T - T // 3, want_planes=False)
    d = b.get_diagnostics()
    li = b.last_launch()
    b.close()
    if fast:
        assert li["kernel"].endswith("true>"), li          # the Full instantiation
    clamp_o = np.array([w.n_clamp_warn for w in want])
    bal_o = np.array([w.n_balance_warn for w in want])
    print(name, "clamp warnings", d["n_clamp_warn"].tolist(), "max|dC| %.2e" % d["max_abs_dC"].max())
    assert clamp_o.sum() > 10000
    np.testing.assert_array_equal(d["n_clamp_warn"], clamp_o)
    np.testing.assert_array_equal(d["n_balance_warn"], bal_o)
    assert d["max_abs_dC"].max() < 1e-9 and (d["max_abs_dC"] > 0).all()
    # setup() zeroes the counters
    b = _batch(flags, clim, members[:4], None, fast=fast, kernel=kernel, options=options, diag=True)
    b.run(0, 100, want_planes=False)
    b.setup()
    assert b.get_diagnostics()["max_abs_dC"].max() == 0.0
    b.close()


@pytest.mark.parametrize("fast,kernel", [(False, sa.KERNEL_AUTO), (True, sa.KERNEL_ONE_WAVE), (True, sa.KERNEL_AUTO)],
                         ids=["strict", "one_wave_ncycle", "auto"])
def test_diagnostics_with_the_nitrogen_cycle_on_the_balance_cases(fast, kernel, oracle):
    """the reference's own balance test data (testBalance.c): carbon AND nitrogen residuals (auto: the cooperative kernels
    where the flag set has the nitrogen cycle -- the soil wave runs the check -- or the optional-physics builds)"""
    from tests.test_balance import CONFIGS, load
    for name in sorted(CONFIGS):
        flags, params, clim, events = load(name)
        st, _, want = oracle.run_member(flags, params, clim, events, want_rec=False)
        assert st == 0
        b = _batch(flags, clim, params[None, :], events, fast=fast, kernel=kernel, diag=True)
        b.run(want_planes=False)
        d = b.get_diagnostics()
        if fast and kernel == sa.KERNEL_AUTO:
            assert b.last_launch()["kernel"].startswith("stepCoop"), b.last_launch()
        b.close()
        assert d["n_clamp_warn"][0] == want.n_clamp_warn and d["n_balance_warn"][0] == want.n_balance_warn == 0, name
        assert d["max_abs_dC"][0] < 1e-8 and d["max_abs_dN"][0] < 1e-8, (name, d)


def test_balance_warnings_are_counted_when_rounding_exceeds_the_threshold(oracle, base):
    """a stand with 1e11 gC of wood: one ulp of the carbon total is 1.5e-5 > EPS = 1e-8, so the
    per-step residual trips checkBalance() on most steps -- in the oracle and in every kernel
    (counts differ by the steps where different rounding lands on either side of the threshold)"""
    from sipnet_amd.config import param_index as pi
    flags = sa.flags_from()
    clim = year_clim(n=48 * 20, start_day=150)
    members = synth.perturbed_params(base, 64)
    members[:, pi("plantWoodInit")] = 1e11
    want = np.array([oracle.run_member(flags, members[m], clim, want_rec=False)[2].n_balance_warn for m in range(8)])
    assert (want > clim.n_steps // 4).all()
    for fast, kernel in ((False, sa.KERNEL_AUTO), (True, sa.KERNEL_COOP_LDS), (True, sa.KERNEL_ONE_WAVE)):
        b = _batch(flags, clim, members, None, fast=fast, kernel=kernel, diag=True)
        b.run(want_planes=False)
        got = b.get_diagnostics()["n_balance_warn"][:8]
        b.close()
        print("balance warnings: oracle", want.tolist(), "gpu", got.tolist())
        assert (np.abs(got - want) <= 0.25 * want).all()


@pytest.mark.parametrize("kernel", [sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_PAIR, sa.KERNEL_ONE_WAVE], ids=["x_lds", "x_pair", "one_wave"])
def test_diagnostics_of_the_optional_physics_kernels_match_the_oracle(kernel, oracle, base):
    """clamp and carbon-balance counters with the litter pool, methane, carbon saturation, flooding, growth
    respiration and leaf water on (the optional-physics Full instantiations): equal to the oracle's on the events
    scenario (harvests, a clear-cut, re-planting: the litter pool takes the transfers and the dead stands)"""
    flags = sa.flags_from(**X_FLAGS)
    xbase = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    clim, ev, members = _scenario(xbase, lethal=True)
    members = members[:70]
    want = [oracle.run_member(flags, members[m], clim, ev, want_rec=False)[2] for m in range(70)]
    b = _batch(flags, clim, members, ev, fast=True, kernel=kernel, diag=True)
    b.run(0, 1001, want_planes=False)
    b.run(1001, clim.n_steps - 1001, want_planes=False)
    d = b.get_diagnostics()
    li = b.last_launch()
    b.close()
    assert li["kernel"].endswith("true>") and ("stepCoopX" in li["kernel"] or kernel == sa.KERNEL_ONE_WAVE), li
    np.testing.assert_array_equal(d["n_clamp_warn"], np.array([w.n_clamp_warn for w in want]))
    np.testing.assert_array_equal(d["n_balance_warn"], np.array([w.n_balance_warn for w in want]))
    assert d["max_abs_dC"].max() < 1e-9


@pytest.mark.parametrize("which,kernel,expect", [("nitrogen", sa.KERNEL_COOP_NCYCLE, "stepCoopNFullKernel<double, false>"),
                                                  ("everything", sa.KERNEL_AUTO, "stepCoopNXFullKernel<double, false>"),
                                                  ("nitrogen", sa.KERNEL_COOP_NCYCLE_PAIR, "stepCoopNPairDiagKernel<double, false>"),
                                                  ("everything", sa.KERNEL_COOP_NCYCLE_PAIR, "stepCoopNXPairDiagKernel<double, false>"),
                                                  ("nitrogen", sa.KERNEL_ONE_WAVE, "stepFastKernel<double, false, 2, 1, true>")],
                         ids=["n_full", "nx_full_auto", "n_pair_diag", "nx_pair_diag", "one_wave"])
def test_diagnostics_counters_with_the_nitrogen_cycle_on_the_cooperative_kernels(which, kernel, expect, oracle, base):
    """round 5: clamp and carbon / nitrogen balance counters with the nitrogen cycle from the cooperative kernels -- the
    plant side's mass totals travel with wave C's end-of-step post to the soil wave, which runs checkBalance()
    (balance.c:122-169 over the pools of nitrogen.c:210-239) -- on the events scenario (fertiliser, harvests, a clear-cut,
    re-planting, leaf events): counters equal to the oracle's, residuals at rounding level.  Round 6: the two-chunk layout too --
    one slot of the eleven mass-total rows per chunk, the carbon wave waiting for the soil wave's check of the step before"""
    flags = sa.flags_from(**(EVERYTHING if which == "everything" else dict(litterPool=1, anaerobic=1, nitrogenCycle=1)))
    nbase = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    clim, ev, members = _scenario(nbase, lethal=True)
    members = members[:70]
    want = [oracle.run_member(flags, members[m], clim, ev, want_rec=False)[2] for m in range(70)]
    b = _batch(flags, clim, members, ev, fast=True, kernel=kernel, diag=True)
    b.run(0, 1001, want_planes=False)
    b.run(1001, clim.n_steps - 1001, want_planes=False)
    d = b.get_diagnostics()
    li = b.last_launch()
    b.close()
    assert li["kernel"] == expect, li
    np.testing.assert_array_equal(d["n_clamp_warn"], np.array([w.n_clamp_warn for w in want]))
    np.testing.assert_array_equal(d["n_balance_warn"], np.array([w.n_balance_warn for w in want]))
    # the largest residuals are the oracle's: rounding level for carbon; for nitrogen the reference's own bookkeeping
    # leaves a real residual on some event steps of this scenario (which is what its warning counts) -- the same one here
    np.testing.assert_allclose(d["max_abs_dC"], np.array([w.max_abs_dC for w in want]), rtol=0, atol=1e-9)
    np.testing.assert_allclose(d["max_abs_dN"], np.array([w.max_abs_dN for w in want]), rtol=1e-6, atol=1e-9)
    assert (d["max_abs_dC"] > 0).any() and (d["max_abs_dN"] > 0).any()

