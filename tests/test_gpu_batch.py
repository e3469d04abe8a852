"""GPU tests of the batch engine beyond single-member parity: throughput kernel with events,
chunked (segmented) runs, ragged member counts, several sites, per-member status codes,
site-fatal plan errors, state round trips, fp32-mixed tolerance, ensemble statistics."""
import os

import numpy as np
import pytest
import torch

import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd._lib import SipnetError
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")


def make_batch(flags, clims, members, prec=sa.F64, events=None, fast=True, kernel=sa.KERNEL_AUTO,
               kernel_options=0):
    b = sa.Batch(flags, len(clims), members.shape[0], prec, fast_math=fast, kernel=kernel,
                 kernel_options=kernel_options)
    for s, c in enumerate(clims):
        if events is not None:
            b.set_events(s, events)
        b.set_climate(s, c)
        b.set_params(s, members)
    b.setup()
    return b


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


@pytest.fixture(scope="module")
def short_clim():
    return synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(48 * 40)))


def test_throughput_kernel_with_events_matches_oracle(oracle, tmp_path):
    """russell_1 = default flags + 108 irrigation events: the lean launch takes the fast
    kernel's event path; perturbed members exercise per-member arithmetic."""
    case = helpers.load_smoke_case("russell_1", str(tmp_path))
    members = synth.perturbed_params(case["params"], 70, names={"aMax", "soilWHC", "wueConst", "baseVegResp"})
    b = make_batch(case["flags"], [case["clim"]], members, events=case["events"])
    planes, _ = b.run()
    got = planes.cpu().numpy()
    b.close()
    want, _, st = oracle.run_block(case["flags"], members, case["clim"], case["events"])
    assert (st == 0).all()
    print("events, fast kernel: max|d|", np.abs(got - want).max())
    assert np.abs(got - want).max() < 1e-9


def test_other_event_types_on_both_kernels(oracle, base, short_clim):
    """plant / harvest / fertilise / till / canopy irrigation on the default-flag model."""
    ev = []
    def add(day, typ, *p):
        e = sa.Event(); e.type = typ; e.year = int(short_clim.year[0]); e.day = day
        for i, v in enumerate(p): e.p[i] = v
        ev.append(e)
    add(3, 2, 1.5, 0)                      # canopy irrigation
    add(5, 4, 0.4)                         # tillage
    add(9, 1, 0.3, 0.2, 0.1, 0.1)          # harvest
    add(9, 0, 2.0, 30.0, 1.0)              # fertiliser on the same day
    add(20, 3, 40.0, 300.0, 50.0, 60.0)    # planting
    add(30, 1, 1.0, 1.0, 0.0, 0.0)         # clear-cut: plant death
    add(35, 3, 40.0, 300.0, 50.0, 60.0)    # replanting: revival with a fresh ring
    members = synth.perturbed_params(base, 64)
    flags = sa.flags_from()
    want, final_o, st = oracle.run_block(flags, members, short_clim, ev)
    assert (st == 0).all()
    for fast in (False, True):
        b = make_batch(flags, [short_clim], members, events=ev, fast=fast)
        planes, _ = b.run()
        got = planes.cpu().numpy()
        state = b.get_state()
        b.close()
        d = np.abs(got - want).max()
        print(f"fast={fast}: max|d| {d:.3e}; died at", state[0, 30])
        assert d < 1e-9
        assert state[0, 30] >= 0               # the clear-cut killed the stand ...
        assert state[0, 0] > 100.0             # ... and the replanting brought wood back


def test_segmented_run_equals_continuous(base, short_clim):
    """Same guarantee as the reference's testSegmentedEquivalence (testRestartMVP.c:253-303):
    stopping at any step boundary and continuing from the state in HBM changes nothing."""
    members = synth.perturbed_params(base, 100)
    for fast in (False, True):
        b = make_batch(sa.flags_from(), [short_clim], members, fast=fast)
        whole, _ = b.run()
        whole = whole.cpu().numpy()
        s_whole = b.get_state()
        b.setup()
        parts = []
        for a, n in ((0, 7), (7, 500), (507, 1), (508, short_clim.n_steps - 508)):
            p, _ = b.run(a, n)
            parts.append(p.cpu().numpy())
        seg = np.concatenate(parts, axis=1)
        s_seg = b.get_state()
        b.close()
        assert np.array_equal(seg, whole), f"fast={fast}"
        assert np.array_equal(s_seg, s_whole)


def test_ragged_members_and_multiple_sites(oracle, base):
    """Member counts that do not fill a wavefront, and 3 / 8 sites (8 takes the XCD-grouped
    block mapping) with different forcing per site."""
    flags = sa.flags_from()
    for n_sites, n_mem in ((3, 37), (8, 65), (1, 1)):
        clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(48 * 12, site=s)))
                 for s in range(n_sites)]
        members = synth.perturbed_params(base, n_mem)
        b = make_batch(flags, clims, members)
        planes, _ = b.run()
        got = planes.cpu().numpy().reshape(3, -1, n_sites, n_mem)
        b.close()
        for s in range(n_sites):
            want, _, _ = oracle.run_block(flags, members, clims[s])
            assert np.abs(got[:, :, s, :] - want).max() < 1e-9, (n_sites, n_mem, s)


def test_bad_member_gets_status_instead_of_killing_the_batch(oracle, base, short_clim):
    members = synth.perturbed_params(base, 10)
    members[4, sa.config.param_index("leafAllocation")] = 0.8
    members[4, sa.config.param_index("woodAllocation")] = 0.5     # sum > 1: sipnet.c:1117-1122
    b = make_batch(sa.flags_from(), [short_clim], members)
    planes, _ = b.run()
    st = b.get_status()
    got = planes.cpu().numpy()
    b.close()
    assert list(st) == [0, 0, 0, 0, 3, 0, 0, 0, 0, 0]
    want, _, so = oracle.run_block(sa.flags_from(), members, short_clim)
    assert so[4] == 3
    ok = [m for m in range(10) if m != 4]
    assert np.abs(got[:, :, ok] - want[:, :, ok]).max() < 1e-9


def test_site_fatal_conditions_return_reference_exit_codes(base):
    flags = sa.flags_from()
    raw = synth.round_like_file(synth.half_hourly_year_raw(48 * 8))
    # steps shorter than 28.8 min overflow the 250-slot running mean (sipnet.c:1562-1569 -> 7)
    r2 = dict(raw); r2["length"] = np.full(len(raw["year"]), -600.0)
    b = sa.Batch(flags, 1, 4)
    b.set_climate(0, synth.convert_raw(r2)); b.set_params(0, base)
    with pytest.raises(SipnetError) as e:
        b.setup()
    assert e.value.code == 7
    b.close()
    # non-positive step length (events.c:460-465 -> 3)
    c = synth.convert_raw(raw); c.data[5, 0] = 0.0
    b = sa.Batch(flags, 1, 4); b.set_climate(0, c); b.set_params(0, base)
    with pytest.raises(SipnetError) as e:
        b.setup()
    assert e.value.code == 3
    b.close()
    # an event before the first climate record (frontend.c:216-223 -> 5)
    ev = sa.Event(); ev.type = 2; ev.year = int(raw["year"][0]) - 1; ev.day = 200; ev.p[0] = 1.0
    b = sa.Batch(flags, 1, 4); b.set_events(0, [ev]); b.set_climate(0, synth.convert_raw(raw)); b.set_params(0, base)
    with pytest.raises(SipnetError) as e:
        b.setup()
    assert e.value.code == 5
    b.close()


def test_state_round_trip_and_transplant(base, short_clim):
    """checkpoint / restore (state vector + ring block = what restart.c persists per member):
    rewinding a batch, or moving the checkpoint into a fresh batch, continues identically."""
    members = synth.perturbed_params(base, 33)
    b1 = make_batch(sa.flags_from(), [short_clim], members, fast=False)
    b1.run(0, 600, want_planes=False)
    ck = b1.checkpoint()
    rest1, _ = b1.run(600, 300)
    b1.restore(ck)                            # rewind
    rest2, _ = b1.run(600, 300)
    assert torch.equal(rest1, rest2)
    b2 = make_batch(sa.flags_from(), [short_clim], members, fast=True)
    b2.restore(ck)                            # transplant into another batch / kernel
    rest3, _ = b2.run(600, 300)
    assert (rest3 - rest1).abs().max().item() < 1e-9
    b1.close(); b2.close()


def test_fp32_mixed_checkpoint_round_trip_with_its_fp32_ring(base, short_clim):
    """an fp32-mixed batch keeps its running-mean ring in fp32 on the device (the values are fp32 numbers there);
    get_rings / set_rings speak doubles: a checkpoint taken and restored -- in place, and into another batch
    and another cooperative layout -- continues bit for bit"""
    members = synth.perturbed_params(base, 130)
    b1 = make_batch(sa.flags_from(), [short_clim], members, prec=sa.F32_MIXED)
    b1.run(0, 600, want_planes=False)
    ck = b1.checkpoint()
    assert np.array_equal(ck[1], ck[1].astype(np.float32).astype(np.float64))     # fp32 numbers
    rest1, _ = b1.run(600, 300)
    b1.restore(ck)
    rest2, _ = b1.run(600, 300)
    assert torch.equal(rest1, rest2)
    b2 = make_batch(sa.flags_from(), [short_clim], members, prec=sa.F32_MIXED, kernel=sa.KERNEL_COOP_PAIR)
    b2.restore(ck)
    rest3, _ = b2.run(600, 300)
    assert b2.last_launch()["kernel"].startswith("stepCoopPairKernel<float")
    assert torch.equal(rest1, rest3)
    b1.close(); b2.close()


def test_fp32_mixed_tolerance(oracle, base):
    """SIPNET_F32_MIXED: fluxes in fp32, pools fp64.  Stated bound: |dNEE| < 2e-6 gC m-2 per
    half-hourly step, yearly NEE sum within 0.5 gC m-2 of the fp64 oracle."""
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(17520)))
    members = synth.perturbed_params(base, 64)
    b = make_batch(sa.flags_from(), [clim], members, prec=sa.F32_MIXED)
    planes, _ = b.run()
    assert planes.dtype == torch.float32
    got = planes.double().cpu().numpy()
    b.close()
    want, _, _ = oracle.run_block(sa.flags_from(), members, clim)
    d = np.abs(got - want)
    print("fp32-mixed: max|dNEE| %.3e  max|dGPP| %.3e  max|dET| %.3e  max|d sum NEE| %.3e" % (
        d[0].max(), d[1].max(), d[2].max(), np.abs(got[0].sum(0) - want[0].sum(0)).max()))
    assert d[0].max() < 2e-6 and d[1].max() < 2e-6 and d[2].max() < 2e-6
    assert np.abs(got[0].sum(0) - want[0].sum(0)).max() < 0.5


def test_reduce_plane_ensemble_statistics(base, short_clim):
    members = synth.perturbed_params(base, 200)
    clims = [short_clim, short_clim.slice(0, short_clim.n_steps)]
    b = make_batch(sa.flags_from(), clims, members)
    planes, _ = b.run()
    stats = b.reduce_plane(planes[0]).cpu().numpy()
    p = planes[0].cpu().numpy().reshape(-1, 2, 200)
    b.close()
    assert np.allclose(stats[:, :, 0], p.sum(-1), rtol=1e-12, atol=1e-12)
    assert np.allclose(stats[:, :, 1], (p * p).sum(-1), rtol=1e-12, atol=1e-12)
    # odd member counts: rows that are not 16-byte aligned and scalar tails; fp32 planes
    for prec in (sa.F64, sa.F32_MIXED):
        members = synth.perturbed_params(base, 67)
        b = make_batch(sa.flags_from(), [short_clim, short_clim, short_clim], members, prec=prec)
        planes, _ = b.run(0, 50)
        stats = b.reduce_plane(planes[1]).cpu().numpy()
        p = planes[1].double().cpu().numpy().reshape(50, 3, 67)
        b.close()
        assert np.allclose(stats[:, :, 0], p.sum(-1), rtol=1e-12, atol=1e-12)
        assert np.allclose(stats[:, :, 1], (p * p).sum(-1), rtol=1e-12, atol=1e-12)


def test_full_size_properties_10k_members(base):
    """BASELINE size (10240 members x 17520 steps): checks that need no oracle run -- member 0
    is the unperturbed base and must equal a 1-member batch bit for bit; duplicated parameter
    rows give duplicated outputs wherever they sit in a wavefront; NEE sums are finite."""
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(17520)))
    members = synth.perturbed_params(base, 10240)
    members[5000] = members[77]
    members[10239] = members[64]
    b = make_batch(sa.flags_from(), [clim], members)
    s0 = b.get_state()
    planes, _ = b.run()
    s1 = b.get_state()
    # carbon conservation over the whole year for every member (the property behind the
    # reference's per-step balance check, balance.c:122-169): the change of the summed carbon
    # pools equals minus the accumulated NEE
    def total_c(s):   # wood + leaf + soil + coarse + fine roots + accounting delta
        return s[:, 0] + s[:, 1] + s[:, 2] + s[:, 6] + s[:, 7] + s[:, 12]
    resid = (total_c(s1) - total_c(s0)) + s1[:, 19]
    print("carbon balance residual over 17520 steps: max %.3e gC m-2" % np.abs(resid).max())
    assert np.abs(resid).max() < 1e-6
    assert np.allclose(s1[:, 19], planes[0].double().sum(0).cpu().numpy(), rtol=0, atol=1e-8)
    b1 = make_batch(sa.flags_from(), [clim], members[:1])
    p1, _ = b1.run()
    assert torch.equal(planes[:, :, 0], p1[:, :, 0])
    assert torch.equal(planes[:, :, 5000], planes[:, :, 77])
    assert torch.equal(planes[:, :, 10239], planes[:, :, 64])
    assert bool(torch.isfinite(planes).all())
    b.close(); b1.close()


def test_cooperative_and_one_wave_kernels_agree(base, tmp_path):
    """the engine picks the three-wavefront kernel (step_coop.hip) for batches of at most one
    64-member chunk per CU and the one-wave kernel (step_fast.hip) above; both must give the same
    trajectories -- including the rare paths: clear-cut + death, re-planting, irrigation, a member
    dead from the start, ragged last chunk, a launch split at arbitrary steps"""
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(48 * 60)))
    ev = []
    def add(day, typ, *p):
        e = sa.Event(); e.type = typ; e.year = int(clim.year[0]); e.day = day
        for i, v in enumerate(p): e.p[i] = v
        ev.append(e)
    add(4, 2, 1.5, 0)
    add(9, 1, 0.3, 0.2, 0.1, 0.1)
    add(20, 1, 1.0, 1.0, 0.0, 0.0)         # clear-cut: every member dies
    add(30, 3, 40.0, 300.0, 50.0, 60.0)    # re-planting
    add(41, 2, 2.0, 1)
    members = synth.perturbed_params(base, 150)     # 3 chunks, the last one ragged
    from sipnet_amd.config import param_index as pi
    members[5, pi("plantWoodInit")] = 0.0           # never alive, keeps its leaves
    out = {}
    for prec in (sa.F64, sa.F32_MIXED):
        for coop, kern in (("1", sa.KERNEL_COOP_LDS), ("2", sa.KERNEL_COOP_HBM), ("3", sa.KERNEL_COOP_PAIR),
                           ("4", sa.KERNEL_COOP_QUAD), ("0", sa.KERNEL_ONE_WAVE)):
            b = make_batch(sa.flags_from(), [clim], members, prec=prec, events=ev, kernel=kern)
            T = clim.n_steps
            planes, _ = b.alloc_outputs(T)
            for a, z in ((0, 7), (7, 1000), (1000, 1015), (1015, T)):   # odd cuts, a one-step tile tail
                b.run(a, z - a, planes=planes[:, a:z])
            assert b.last_launch()["kernel"].startswith("stepFastKernel" if coop == "0" else "stepCoopPair" if coop == "3" else "stepCoopQuad" if coop == "4" else "stepCoopKernel<")
            out[coop] = (planes.cpu().numpy().astype(np.float64), b.get_state(), b.get_rings())
            b.close()
        tol = 1e-12 if prec == sa.F64 else 2e-4
        for mode in ("1", "2", "3", "4"):
            d = np.abs(out[mode][0] - out["0"][0]).max()
            print("precision", prec, "coop mode", mode, "vs one-wave: max|d|", d)
            assert d < tol
            np.testing.assert_allclose(out[mode][1][:, :27], out["0"][1][:, :27], rtol=1e-9 if prec == sa.F64 else 1e-3, atol=1e-9)
            assert (out[mode][1][:, 28:31] == out["0"][1][:, 28:31]).all()      # ring epoch, status, died-at
            np.testing.assert_allclose(out[mode][2], out["0"][2], rtol=1e-9 if prec == sa.F64 else 1e-3, atol=1e-9)
        np.testing.assert_array_equal(out["1"][0], out["2"][0])   # ring placement does not change arithmetic
        np.testing.assert_array_equal(out["1"][0], out["3"][0])   # nor do the multi-chunk workgroups
        np.testing.assert_array_equal(out["1"][0], out["4"][0])


def test_math_policy_is_an_explicit_choice(base, short_clim, monkeypatch):
    """sipnet_batch_set_math: strict reference order vs throughput kernels on the same batch;
    no environment variable steers the launch path; fp32-mixed has no strict mode."""
    members = synth.perturbed_params(base, 64)
    for k, v in (("SIPNET_FAST_MATH", "1"), ("SIPNET_COOP", "0"), ("SIPNET_NO_FAST_KERNEL", "1")):
        monkeypatch.setenv(k, v)            # round-1 switches: must be ignored now
    b = sa.Batch(sa.flags_from(), 1, 64)
    b.set_climate(0, short_clim)
    b.set_params(0, members)
    b.setup()
    strict = b.run()[0].cpu().numpy()                  # default: strict
    assert b.last_launch()["kernel"] == "stepKernel<Cfg<double, false, false, false>>"
    b.set_math(True)
    b.setup()
    fast = b.run()[0].cpu().numpy()
    assert b.last_launch()["kernel"] == "stepCoopKernel<double, true, true, false>"
    b.set_math(False)
    b.setup()
    strict2 = b.run()[0].cpu().numpy()
    b.close()
    assert np.array_equal(strict, strict2)
    d = np.abs(strict - fast).max()
    assert 0 < d < 1e-12                                # different instruction streams, same model
    # a forced kernel that cannot run the batch is refused, not silently replaced
    b = sa.Batch(sa.flags_from(), 1, 64, kernel=sa.KERNEL_COOP_LDS)
    b.set_climate(0, short_clim)
    b.set_params(0, members)
    b.setup()
    with pytest.raises(SipnetError):
        b.run()                                          # throughput kernel under strict math
    b.set_math(True)
    assert np.array_equal(b.run()[0].cpu().numpy(), fast)
    b.close()
    b32 = sa.Batch(sa.flags_from(), 1, 64, sa.F32_MIXED)
    with pytest.raises(SipnetError):
        b32.set_math(False)
    b32.close()


@pytest.mark.parametrize("which", ["f64_default_flags", "f32_runtime_flags"])
def test_two_waves_per_simd_build_of_the_one_wave_kernel(which, oracle, base):
    """More chunks than SIMDs (> 65 536 members): the launcher takes the instantiation cut to
    256 VGPRs (two resident wavefronts per SIMD, a few spilled registers) of the fp64 default-flag
    kernel and of the fp32 run-time-flag kernel.  Same arithmetic: the planes must equal the
    512-VGPR build's bit for bit, and sampled members the oracle's."""
    M, T = 66_048, 48 * 12
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
    if which == "f64_default_flags":
        flags, prec, tol = sa.flags_from(), sa.F64, 1e-12
        members = synth.perturbed_params(base, M)
    else:
        flags, prec, tol = sa.flags_from(litterPool=1, growthResp=1, leafWater=1), sa.F32_MIXED, 2e-6
        allp = sa.read_params(os.path.join(helpers.GOLDEN, "synth", "allflags.param"), flags)[0]
        members = synth.perturbed_params(allp, M)
    outs = []
    for occ1 in (False, True):
        b = make_batch(flags, [clim], members, prec=prec,
                       kernel_options=sa.KOPT_ONE_WAVE_PER_SIMD if occ1 else 0)
        outs.append(b.run()[0].double().cpu().numpy())
        assert b.last_launch()["kernel"].endswith(", 1, false>" if occ1 else ", 2, false>"), b.last_launch()
        b.close()
    assert np.array_equal(outs[0], outs[1])
    pick = np.r_[0:16, M // 2:M // 2 + 16, M - 16:M]
    want, _, st = oracle.run_block(flags, members[pick], clim)
    assert (st == 0).all()
    assert np.abs(outs[0][:, :, pick] - want).max() < tol


STAT_KERNELS = [("coop_lds", sa.KERNEL_COOP_LDS), ("coop_hbm", sa.KERNEL_COOP_HBM), ("coop_pair", sa.KERNEL_COOP_PAIR),
                ("coop_quad", sa.KERNEL_COOP_QUAD), ("one_wave", sa.KERNEL_ONE_WAVE), ("strict", sa.KERNEL_STRICT)]


@pytest.mark.parametrize("name,kernel", STAT_KERNELS, ids=[k[0] for k in STAT_KERNELS])
@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_run_stats_equals_the_sums_over_the_planes(name, kernel, prec, base, short_clim):
    """sipnet_batch_run_stats: the statistics block of a launch -- summed by the cooperative kernels'
    light wave from the freshly stored plane tiles, or by three reduction passes behind the other
    kernels -- equals the sums over the very planes the launch wrote.  Ragged chunks (67 and 200
    members), three sites (odd number of chunks per workgroup pair / quad), launches cut at steps
    that are not tile boundaries, a one-step launch."""
    if prec == sa.F32_MIXED and kernel == sa.KERNEL_STRICT:
        pytest.skip("fp32-mixed has no strict kernel")
    T = 48 * 12 + 5
    clim = short_clim.slice(0, T)
    for M in (67, 200):
        members = synth.perturbed_params(base, M)
        # (KOPT_STATS_IN_KERNEL: the light wave sums on the two- / four-chunk layouts as well, where
        # the default is the three reduction passes)
        b = make_batch(sa.flags_from(), [clim, clim.slice(0, T), clim.slice(0, T)], members, prec=prec,
                       fast=kernel != sa.KERNEL_STRICT, kernel=kernel,
                       kernel_options=sa.KOPT_STATS_IN_KERNEL if name.startswith("coop") else 0)
        planes, _ = b.alloc_outputs(T)
        stats = torch.full((3, T, 3, 2), float("nan"), dtype=torch.float64, device=planes.device)
        for a0, z in ((0, 1), (1, 23), (23, 400), (400, T)):
            _, st = b.run_stats(a0, z - a0, planes=planes[:, a0:z])
            stats[:, a0:z] = st
        li = b.last_launch()
        p = planes.double().cpu().numpy().reshape(3, T, 3, M)
        got = stats.cpu().numpy()
        # the same planes as a plain run writes
        b.setup()
        ref, _ = b.run()
        same = bool((ref == planes).all())
        b.close()
        assert same, li
        scale = np.abs(p).sum(-1).max()
        assert np.isfinite(got).all(), li
        assert np.abs(got[..., 0] - p.sum(-1)).max() < 1e-12 * max(scale, 1.0), li
        assert np.abs(got[..., 1] - (p * p).sum(-1)).max() < 1e-12 * max((p * p).sum(-1).max(), 1.0), li


@pytest.mark.parametrize("name,fast,kernel", [("strict", False, sa.KERNEL_AUTO), ("one_wave", True, sa.KERNEL_ONE_WAVE),
                                              ("coop_lds", True, sa.KERNEL_COOP_LDS), ("coop_hbm", True, sa.KERNEL_COOP_HBM),
                                              ("coop_pair", True, sa.KERNEL_COOP_PAIR), ("coop_quad", True, sa.KERNEL_COOP_QUAD)])
def test_sites_of_different_lengths_in_one_batch(name, fast, kernel):
    """three sites whose forcings have 4 800, 4 023 and 2 400 records in ONE batch (a launch advances every site to
    the end of its own records; launches cut at odd steps, one of them across a site's end): planes, statistics and
    final state equal three one-site batches, bit for bit; rows past a site's end are left untouched"""
    flags = sa.flags_from()
    base = sa.read_params(os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param"), flags)[0]
    lens = [4800, 4023, 2400]
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(n, site=s))) for s, n in enumerate(lens)]
    M = 130
    members = synth.perturbed_params(base, M)
    b = sa.Batch(flags, 3, M, sa.F64, fast_math=fast, kernel=kernel)
    for s in range(3):
        b.set_climate(s, clims[s])
        b.set_params(s, members)
    b.setup()
    assert b.n_steps == 4800 and [b.site_n_steps(s) for s in range(3)] == lens
    T = 4800
    planes = torch.full((3, T, 3 * M), -7.0, dtype=torch.float64, device="cuda")
    stats = torch.zeros((3, T, 3, 2), dtype=torch.float64, device="cuda")
    cuts = [0, 7, 2399, 2405, 4030, T]
    for a, z in zip(cuts[:-1], cuts[1:]):
        _, seg = b.run_stats(a, z - a, planes=planes[:, a:z])
        stats[:, a:z] = seg
    got, state, st = planes.cpu().numpy().reshape(3, T, 3, M), b.get_state().reshape(3, M, -1), stats.cpu().numpy()
    li = b.last_launch()["kernel"]
    g, _ = b.site_series(1)
    b.close()
    for s in range(3):
        one = sa.Batch(flags, 1, M, sa.F64, fast_math=fast, kernel=kernel)
        one.set_climate(0, clims[s])
        one.set_params(0, members)
        one.setup()
        p1, s1 = one.run_stats(0, lens[s])
        assert one.last_launch()["kernel"] == li
        np.testing.assert_array_equal(got[:, :lens[s], s], p1.cpu().numpy(), err_msg=f"site {s}")
        np.testing.assert_array_equal(state[s][:, :28], one.get_state()[:, :28], err_msg=f"site {s}")
        np.testing.assert_allclose(st[:, :lens[s], s], s1.cpu().numpy()[:, :, 0], rtol=1e-12, atol=1e-12)
        assert (got[:, lens[s]:, s] == -7.0).all()                     # rows past the site's end: untouched
        one.close()
