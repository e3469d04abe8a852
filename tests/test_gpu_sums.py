"""sipnet_batch_run_sums: every member's sums of NEE / GPP / ET over groups of k consecutive steps, accumulated inside the
step kernel's own launch by the wavefront that computes the value (step_coop.hip, coopBody<..., Sums>: stepCoopSumsKernel,
stepCoopPairSumsKernel; step_coop_sums.hip: fp32-mixed batches and the four-chunk layout) -- against the SAME batch's per-step planes summed on the host in step order: bit for bit, on every
layout that has such a kernel, with split launches, groups that do not divide the run, ragged chunks, sites of different
lengths, members with a general VPD exponent, a member that dies and one that never runs; and against the CPU oracle's
daily sums (1e-9).  What a consumer of the reference's per-step output rows (sipnet.c:453-473) aggregates anyway."""
import os

import numpy as np
import pytest
import torch

import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


def host_sums(planes, k):
    """planes [3][T][ncol] -> [3][ceil(T / k)][ncol], added in step order like the kernel"""
    T = planes.shape[1]
    out = np.zeros((3, (T + k - 1) // k, planes.shape[2]))
    for t in range(T):
        out[:, t // k] += planes[:, t]
    return out


@pytest.mark.parametrize("kernel,name", [(sa.KERNEL_COOP_LDS, "stepCoopSumsKernel<true, true>"), (sa.KERNEL_COOP_HBM, "stepCoopSumsKernel<true, false>"),
                                         (sa.KERNEL_COOP_PAIR, "stepCoopPairSumsKernel<true>"), (sa.KERNEL_AUTO, "stepCoopSumsKernel<true, true>")],
                         ids=["lds-ring", "hbm-ring", "pair", "auto"])
def test_sums_in_the_launch_equal_the_planes_summed_in_step_order(base, kernel, name):
    M, T = 64 * 3 + 17, 48 * 5 + 11
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(3)]
    clims[1] = clims[1].slice(0, T - 29)                     # a site that ends inside a group
    members = synth.perturbed_params(base, M, seed=3)
    members[5, pi("leafAllocation")] = 0.9                   # never runs (status 3)
    members[5, pi("woodAllocation")] = 0.9
    members[9, pi("plantWoodInit")] = 1e-4                   # starves: dies inside the run
    members[9, pi("laiInit")] = 1e-4

    def make():
        b = sa.Batch(sa.flags_from(), 3, M, sa.F64, fast_math=True, kernel=kernel)
        b.set_climates(clims)
        b.set_params(None, members)
        b.setup()
        return b

    ref = make()
    planes, _ = ref.run(0, T)
    want = planes.cpu().numpy()
    want[:, T - 29:, M:2 * M] = 0.0                          # (rows past the shorter site's end: never written)
    ran = np.tile(ref.get_status()[:M] == 0, 3)              # (what a member that never runs leaves in a plane means nothing, and
    assert (~ran).sum() == 3                                 #  depends on where the launches were cut: left out below)
    ref.close()
    for k, cuts in ((48, [0, T]), (48, [0, 96, T]), (7, [0, 7 * 9, 7 * 20, T]), (T + 5, [0, T]), (1, [0, 33, T])):
        b = make()
        assert b.sums_in_kernel()
        got = []
        for a, z in zip(cuts[:-1], cuts[1:]):
            assert a % k == 0                                # launches start at group boundaries
            got.append(b.run_sums(a, z - a, k).cpu().numpy())
            assert b.last_launch()["kernel"] == name, b.last_launch()["kernel"]
        got = np.concatenate(got, axis=1)
        ws = host_sums(want, k)
        # (the shorter site's groups past its end: what a launch does not reach is not written -- compare what is)
        g_end = (T - 29 + k - 1) // k
        got, ws = got[:, :, ran], ws[:, :, ran]
        Mr = M - 1
        np.testing.assert_array_equal(got[:, :, :Mr], ws[:, :, :Mr])
        np.testing.assert_array_equal(got[:, :, 2 * Mr:], ws[:, :, 2 * Mr:])
        np.testing.assert_array_equal(got[:, :g_end, Mr:2 * Mr], ws[:, :g_end, Mr:2 * Mr])
        # the state after a summing run is the state after the plain run
        b2 = make()
        b2.run(0, T)
        np.testing.assert_array_equal(b.get_state(), b2.get_state())
        b.close()
        b2.close()


def test_daily_sums_against_the_oracle_and_general_exponents(base):
    M, T, K = 130, 48 * 4, 48
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
    members = synth.perturbed_params(base, M, seed=8)
    members[3, pi("dVpdExp")] = 1.7                          # the general-exponent instantiation
    b = sa.Batch(sa.flags_from(), 1, M, sa.F64, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    got = b.run_sums(0, T, K).cpu().numpy()
    assert b.last_launch()["kernel"] == "stepCoopSumsKernel<false, true>"
    b.close()
    ora = helpers.load_oracle()
    ref, _, st = ora.run_block(sa.flags_from(), members[:8], clim)
    assert (st == 0).all()
    ref_sums = np.stack([ref[:, g * K:(g + 1) * K].sum(axis=1) for g in range(T // K)], axis=1)
    np.testing.assert_allclose(got[:, :, :8], ref_sums, rtol=0, atol=1e-9)


NCYC = dict(litterPool=1, anaerobic=1, nitrogenCycle=1)
FLAG_SETS = {"russell_3": dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0), "anaerobic_litter": dict(anaerobic=1, litterPool=1),
             "nitrogen": NCYC, "everything": dict(carbonSaturation=1, flooding=1, growthResp=1, leafWater=1, **NCYC)}


@pytest.mark.parametrize("which", sorted(FLAG_SETS))
@pytest.mark.parametrize("sites,kernel_prefix", [(1, ""), (4, "Pair")], ids=["one-chunk", "two-chunk"])
def test_sums_of_the_optional_physics_and_nitrogen_cycle_layouts(which, sites, kernel_prefix):
    """the same for every physics family that has cooperative kernels: the optional-physics instantiations (stepCoopXSumsKernel,
    stepCoopXPairSumsKernel) and the nitrogen-cycle layouts, whose SOIL wave forms NEE and therefore sums it (stepCoopNSumsKernel,
    stepCoopNPairSumsKernel; "everything": with the other options on top) -- events, split launches, a group length that does
    not divide the run: bit for bit against the same batch's planes added up in step order"""
    import ctypes as C
    from sipnet_amd._lib import Event
    flags = sa.flags_from(**FLAG_SETS[which])
    base = sa.read_params(os.path.join(helpers.REPO, "sipnet_amd", "data", "allflags_forest.param"), flags)[0]
    M, T = 64 * 2 + 9, 48 * 6 + 5
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(sites)]
    members = synth.perturbed_params(base, M, seed=13)
    ev = [Event(type=t, year=int(clims[0].year[48 * d]), day=int(clims[0].day[48 * d]), pad=0, p=(C.c_double * 4)(*p))
          for t, d, p in ((4, 1, (0.5, 0, 0, 0)), (2, 2, (1.5, 1.0, 0, 0)), (0, 3, (1.0, 5.0, 2.0, 0.0)))]       # tillage, irrigation, fertiliser
    ncyc = "nitrogenCycle" in FLAG_SETS[which]
    kernel = (sa.KERNEL_COOP_NCYCLE_PAIR if ncyc else sa.KERNEL_COOP_PAIR) if kernel_prefix else sa.KERNEL_AUTO

    def make():
        b = sa.Batch(flags, sites, M, sa.F64, fast_math=True, kernel=kernel)
        for s_ in range(sites):
            b.set_events(s_, ev)
        b.set_climates(clims)
        b.set_params(None, members)
        b.setup()
        return b

    ref = make()
    planes, _ = ref.run(0, T)
    want = planes.cpu().numpy()
    assert (ref.get_status() == 0).all()
    state = ref.get_state()
    ref.close()
    for k, cuts in ((48, [0, 96, T]), (7, [0, 70, T])):
        b = make()
        assert b.sums_in_kernel()
        got = np.concatenate([b.run_sums(a, z - a, k).cpu().numpy() for a, z in zip(cuts[:-1], cuts[1:])], axis=1)
        name = b.last_launch()["kernel"]
        assert name.startswith("stepCoop%s%sSumsKernel" % ("N" if ncyc else "X", kernel_prefix)), name
        np.testing.assert_array_equal(got, host_sums(want, k))
        np.testing.assert_array_equal(b.get_state(), state)
        b.close()


LAYOUTS = {"lds": (sa.KERNEL_COOP_LDS, 0), "hbm": (sa.KERNEL_COOP_HBM, 1), "pair": (sa.KERNEL_COOP_PAIR, 2), "quad": (sa.KERNEL_COOP_QUAD, 3),
           "n": (sa.KERNEL_COOP_NCYCLE, 4), "npair": (sa.KERNEL_COOP_NCYCLE_PAIR, 5)}


@pytest.mark.parametrize("prec,layout,which", [
    (sa.F32_MIXED, "lds", None), (sa.F32_MIXED, "hbm", None), (sa.F32_MIXED, "pair", None), (sa.F32_MIXED, "quad", None), (sa.F64, "quad", None),
    (sa.F32_MIXED, "lds", "russell_3"), (sa.F32_MIXED, "pair", "anaerobic_litter"), (sa.F32_MIXED, "quad", "russell_3"),
    (sa.F32_MIXED, "n", "nitrogen"), (sa.F32_MIXED, "npair", "nitrogen"), (sa.F32_MIXED, "n", "everything"), (sa.F32_MIXED, "npair", "everything")],
    ids=lambda v: v if isinstance(v, str) else ("default" if v is None else ("f64" if v == sa.F64 else "f32mixed")))
def test_sums_of_fp32_mixed_batches_and_of_the_four_chunk_layout(prec, layout, which):
    """step_coop_sums.hip (stepCoopSumsAtKernel): the in-launch sums of fp32-mixed batches on every layout and family and of fp64
    batches on the four-chunk layout.  The sums are DOUBLES whatever the arithmetic type: a float value is widened and added in step
    order -- bit for bit the same batch's planes added up that way on the host; ragged chunks, three sites of different lengths,
    split launches, a group length that does not divide the run"""
    flags = sa.flags_from(**(FLAG_SETS[which] if which else {}))
    base = sa.read_params(os.path.join(helpers.REPO, "sipnet_amd", "data", "allflags_forest.param" if which else "base_forest.param"), flags)[0]
    M, T = 64 * 3 + 17, 48 * 5 + 11
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(3)]
    clims[2] = clims[2].slice(0, T - 40)
    members = synth.perturbed_params(base, M, seed=21)
    if not which:
        members[7, pi("dVpdExp")] = 1.6                      # the general-exponent instantiations
    kernel, code = LAYOUTS[layout]

    def make():
        b = sa.Batch(flags, 3, M, prec, fast_math=True if prec == sa.F64 else None, kernel=kernel)
        b.set_climates(clims)
        b.set_params(None, members)
        b.setup()
        return b

    ref = make()
    planes, _ = ref.run(0, T)
    assert (ref.get_status() == 0).all()
    want = planes.double().cpu().numpy()
    want[:, T - 40:, 2 * M:] = 0.0
    state = ref.get_state()
    ref.close()
    for k, cuts in ((48, [0, 96, T]), (7, [0, 7 * 11, T]), (T, [0, T])):
        b = make()
        assert b.sums_in_kernel()
        got = np.concatenate([b.run_sums(a, z - a, k).cpu().numpy() for a, z in zip(cuts[:-1], cuts[1:])], axis=1)
        name = b.last_launch()["kernel"]
        assert name.startswith("stepCoopSumsAtKernel<%s, " % ("double" if prec == sa.F64 else "float")) and name.endswith(
            ", %d, %s>" % (code, "true" if which in ("russell_3", "anaerobic_litter", "everything") else "false")), name
        ws = host_sums(want, k)
        g_end = (T - 40 + k - 1) // k
        np.testing.assert_array_equal(got[:, :, :2 * M], ws[:, :, :2 * M])
        np.testing.assert_array_equal(got[:, :g_end, 2 * M:], ws[:, :g_end, 2 * M:])
        np.testing.assert_array_equal(b.get_state(), state)
        b.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32mixed"])
@pytest.mark.parametrize("which", [None, "russell_3", "nitrogen", "everything"])
def test_sums_of_the_one_wavefront_kernel(prec, which):
    """step_fast_sums.hip (stepFastSumsKernel): batches the shape policy gives the one-wavefront kernel (here: forced) -- the
    default, nitrogen-cycle and run-time-flag instantiations, both register budgets (SIPNET_KOPT_ONE_WAVE_PER_SIMD), ragged chunks,
    sites of different lengths, split launches, an armed particle-filter forecast in the same launch"""
    flags = sa.flags_from(**(FLAG_SETS[which] if which else {}))
    base = sa.read_params(os.path.join(helpers.REPO, "sipnet_amd", "data", "allflags_forest.param" if which else "base_forest.param"), flags)[0]
    M, T = 64 * 2 + 30, 48 * 4 + 13
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(2)]
    clims[1] = clims[1].slice(0, T - 25)
    members = synth.perturbed_params(base, M, seed=31)
    mode = {None: 0, "nitrogen": 2}.get(which, 1)
    for kopt in (0, sa.KOPT_ONE_WAVE_PER_SIMD):
        def make():
            b = sa.Batch(flags, 2, M, prec, fast_math=True if prec == sa.F64 else None, kernel=sa.KERNEL_ONE_WAVE, kernel_options=kopt)
            b.set_climates(clims)
            b.set_params(None, members)
            b.setup()
            return b

        ref = make()
        planes, _ = ref.run(0, T)
        assert (ref.get_status() == 0).all()
        want = planes.double().cpu().numpy()
        want[:, T - 25:, M:] = 0.0
        state = ref.get_state()
        ref.close()
        for k, cuts in ((48, [0, 96, T]), (5, [0, 5 * 13, T])):
            b = make()
            assert b.sums_in_kernel()
            got = np.concatenate([b.run_sums(a, z - a, k).cpu().numpy() for a, z in zip(cuts[:-1], cuts[1:])], axis=1)
            name = b.last_launch()["kernel"]
            assert name.startswith("stepFastSumsKernel<%s, " % ("double" if prec == sa.F64 else "float")) and ", %d, " % mode in name, name
            ws = host_sums(want, k)
            g_end = (T - 25 + k - 1) // k
            # to the arithmetic's last bits, not bit for bit like the cooperative kernels': this build and the one that stores the planes
            # are both compiled with -ffp-contract=fast, and without the plane stores a few products on rare paths fuse differently
            # (seen by the fuzzer once in ~10^6 values: one flux off by its last bit)
            same = (lambda x, y: np.testing.assert_allclose(x, y, rtol=1e-6, atol=1e-8)) if prec == sa.F32_MIXED else \
                (lambda x, y: np.testing.assert_allclose(x, y, rtol=1e-13, atol=1e-16))
            same(got[:, :, :M], ws[:, :, :M])
            same(got[:, :g_end, M:], ws[:, :g_end, M:])
            # the state after a summing launch is the plain launch's to the last bits (seen: fp64's GPP tracker, state column 14 --
            # the plain build fuses the product that is GPP into the tracker's addition)
            st = b.get_state()
            bad = np.argwhere(st != state)
            np.testing.assert_allclose(st, state, rtol=1e-13 if prec == sa.F64 else 1e-5, atol=1e-300 if prec == sa.F64 else 1e-7)
            b.close()


def test_fp32_mixed_daily_sums_against_the_oracle(base):
    """... and against the CPU oracle's daily sums, within what fp32-mixed arithmetic leaves of a day's sum of 48 half-hourly fluxes"""
    M, T, K = 64 * 4 * 4, 48 * 6, 48
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
    members = synth.perturbed_params(base, M, seed=5)
    b = sa.Batch(sa.flags_from(), 1, M, sa.F32_MIXED, kernel=sa.KERNEL_COOP_QUAD)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    got = b.run_sums(0, T, K).cpu().numpy()
    assert b.last_launch()["kernel"] == "stepCoopSumsAtKernel<float, true, 3, false>"
    b.close()
    ora = helpers.load_oracle()
    pick = [0, 63, 64, 500, M - 1]
    ref, _, st = ora.run_block(sa.flags_from(), members[pick], clim)
    assert (st == 0).all()
    ref_sums = np.stack([ref[:, g * K:(g + 1) * K].sum(axis=1) for g in range(T // K)], axis=1)
    np.testing.assert_allclose(got[:, :, pick], ref_sums, rtol=2e-4, atol=2e-4)


def test_batches_without_such_a_kernel_say_so(base):
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(96)))
    for kw in (dict(prec=sa.F64, fast_math=False), dict(prec=sa.F64, fast_math=True, diagnostics=True),
               dict(prec=sa.F64, fast_math=True, kernel_options=sa.KOPT_FULL_STATE)):
        prec = kw.pop("prec")
        flags = kw.pop("flags", sa.flags_from())
        diag = kw.pop("diagnostics", False)
        b = sa.Batch(flags, 1, 64, prec, **kw)
        if diag:
            b.enable_diagnostics()
        b.set_climate(0, clim)
        b.set_params(0, base)
        b.setup()
        assert not b.sums_in_kernel()
        with pytest.raises(sa.SipnetError) as e:
            b.run_sums(0, 96, 48)
        assert "sums" in str(e.value)
        b.close()
