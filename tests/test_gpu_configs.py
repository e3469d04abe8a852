"""Every BASELINE.json configuration at its OWN shape against the CPU oracle, and every
throughput-kernel instantiation against the oracle (not against a sibling kernel).

  C2  1 site x 1024 members, fp64, 17 520 steps       -> every member vs the oracle
  C3  1 site x 65 536 members, fp32-mixed, 17 520     -> 48 sampled members (first / middle / last chunk)
  C4  32 sites x 1024 members, fp64 (one GPU's share of 256 x 1024), 17 520 steps
                                                      -> 2 sampled members of every site
  C5  131 072 particles x 48 steps + one analysis     -> forecast of sampled particles, weights,
      ancestors, resampled state vs their ancestors' oracle trajectories, second forecast vs an
      unfiltered twin
Each test also asserts WHICH kernel instantiation the engine launched (sipnet_batch_last_launch),
so a change of the selection policy cannot silently move a configuration to another kernel.

Tolerances (stated): fp64 1e-9 gC m-2 (cm) per step on NEE / GPP / ET, against a bar of 1e-6;
fp32-mixed 2e-6 per step on flux planes (pools and accumulators stay fp64).  The oracle is
oracle/sipnet_oracle.c, bit-identical to the real reference (tests/test_oracle_golden.py);
the C5 analysis oracle (oracle/pf_oracle.py) is textbook code -- parity unpinned by any
reference, the reference has no filter.
"""
import os

import numpy as np
import pytest
import torch

import sipnet_amd as sa
from oracle import pf_oracle as po
from sipnet_amd import dist as sd
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")
TOL_F64, TOL_F32 = 1e-9, 2e-6
T_YEAR = 17520


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


def year_clim(site=0, n=T_YEAR, start_day=0):
    raw = synth.half_hourly_year_raw(start_day * 48 + n, site=site)
    raw = {k: v[start_day * 48:] for k, v in raw.items()}
    return synth.convert_raw(synth.round_like_file(raw))


def build(flags, clims, members, prec, kernel=sa.KERNEL_AUTO, options=0, events=None):
    b = sa.Batch(flags, len(clims), members.shape[0], prec,
                 fast_math=True if prec == sa.F64 else None, kernel=kernel, kernel_options=options)
    for s, c in enumerate(clims):
        if events is not None:
            b.set_events(s, events)
        b.set_climate(s, c)
        b.set_params(s, members)
    b.setup()
    return b


def branch_flips(got, want, scale_floor=1e-3):
    """members whose trajectory leaves the oracle's by more than 1e-4 of a plane's maximum on any
    step: what a threshold branch taken one step apart (snow gone, soil dry) looks like"""
    scale = np.maximum(np.abs(want).max(axis=(1, 2), keepdims=True), scale_floor)
    return int(((np.abs(got - want) / scale) > 1e-4).any(axis=(0, 1)).sum())


# ---------------------------------------------------------------------------------------------
def test_c2_1024_members_fp64_every_member(oracle, base):
    clim = year_clim()
    members = synth.perturbed_params(base, 1024)
    b = build(sa.flags_from(), [clim], members, sa.F64)
    planes, _ = b.run()
    li = b.last_launch()
    got = planes.cpu().numpy()
    st = b.get_status()
    b.close()
    assert li["kernel"] == "stepCoopKernel<double, true, true, false>" and li["grid"] == 16
    want, _, so = oracle.run_block(sa.flags_from(), members, clim)
    assert (st == 0).all() and (so == 0).all()
    d = np.abs(got - want).max(axis=(1, 2))
    print("C2 1x1024 f64 x 17520: max|dNEE| %.3e |dGPP| %.3e |dET| %.3e, branch-flip members %d"
          % (d[0], d[1], d[2], branch_flips(got, want)))
    assert d.max() < TOL_F64
    # the bound include/sipnet_amd.h documents for SIPNET_MATH_FAST on the benchmark ensemble (degree-9 exp2)
    assert d[0] < 2e-14 and d[1] < 2e-14 and d[2] < 2e-14


def test_c3_65536_members_fp32_mixed(oracle, base):
    """the HBM-roofline configuration: 13.8 GB of fp32 planes stay resident; four chunks per CU run
    as twelve-wave workgroups of the cooperative kernel"""
    M = 65536
    clim = year_clim()
    members = synth.perturbed_params(base, M)
    b = build(sa.flags_from(), [clim], members, sa.F32_MIXED)
    planes, _ = b.run()
    li = b.last_launch()
    assert planes.dtype == torch.float32 and tuple(planes.shape) == (3, T_YEAR, M)
    # the first / middle / last sixteen members and ONE random member of every 64-member chunk (1 024 chunks: the
    # workgroup -> chunk mapping of the four-chunk layout is then checked at the full shape, not only at small ones)
    rng = np.random.default_rng(3)
    pick = np.unique(np.r_[0:16, M // 2:M // 2 + 16, M - 16:M, np.arange(M // 64) * 64 + rng.integers(0, 64, M // 64)])
    got = planes[:, :, torch.from_numpy(pick).to(planes.device)].double().cpu().numpy()
    st = b.get_status()
    finite = bool(torch.isfinite(planes).all())
    state = b.get_state()
    b.close()
    del planes
    torch.cuda.empty_cache()
    assert li["kernel"] == "stepCoopQuadKernel<float, true>" and li["grid"] == 256, li
    assert (st == 0).all() and finite
    want, final, so = oracle.run_block(sa.flags_from(), members[pick], clim)
    assert (so == 0).all()
    d = np.abs(got - want)
    flips = branch_flips(got, want)
    print("C3 1x65536 f32-mixed x 17520, %d sampled members (one of every chunk): max|dNEE| %.3e |dGPP| %.3e |dET| %.3e, "
          "yearly NEE sum off by at most %.3e gC m-2, branch-flip members %d"
          % (len(pick), d[0].max(), d[1].max(), d[2].max(), np.abs(got[0].sum(0) - want[0].sum(0)).max(), flips))
    assert d.max() < TOL_F32
    assert flips == 0
    assert np.abs(got[0].sum(0) - want[0].sum(0)).max() < 0.5
    # pools (fp64 on the device) of the sampled members after the year
    np.testing.assert_allclose(state[pick, :13], final[:, 14:27], rtol=2e-4, atol=1e-3)


def test_c4_32_sites_x_1024_members_fp64(oracle, base):
    """one GPU's share of the 256-site configuration: the paired-chunk cooperative kernel (two
    chunks per workgroup, ring in HBM) with the XCD-grouped block mapping (n_sites % 8 == 0), every
    site with its own forcing"""
    S, M = 32, 1024
    clims = [year_clim(site=s) for s in range(S)]
    members = synth.perturbed_params(base, M)
    b = build(sa.flags_from(), clims, members, sa.F64)
    planes, _ = b.run()
    li = b.last_launch()
    rng = np.random.default_rng(4)
    # one random member of EVERY chunk of every site (512 chunks: the paired, XCD-grouped workgroup -> chunk mapping
    # is checked at the full shape)
    K = M // 64
    pick = np.stack([np.arange(K) * 64 + rng.integers(0, 64, K) for _ in range(S)])      # [S][K]
    cols = (np.arange(S)[:, None] * M + pick).reshape(-1)
    got = planes[:, :, torch.from_numpy(cols).to(planes.device)].cpu().numpy().reshape(3, T_YEAR, S, K)
    st = b.get_status()
    b.close()
    del planes
    torch.cuda.empty_cache()
    assert li["kernel"] == "stepCoopPairKernel<double, true, false>" and li["grid"] == 256, li
    assert li["plan_threads"] >= 1 and li["plan_build_ms"] > 0
    assert (st == 0).all()
    worst = 0.0
    for s in range(S):
        want, _, so = oracle.run_block(sa.flags_from(), members[pick[s]], clims[s])
        assert (so == 0).all()
        worst = max(worst, float(np.abs(got[:, :, s, :] - want).max()))
        assert np.abs(got[:, :, s, :] - want).max() < TOL_F64, s
    print("C4 32x1024 f64 x 17520, one member of every chunk of every site (512): max|d| %.3e; plans built by %d threads in %.1f ms, "
          "uploaded in %.1f ms" % (worst, li["plan_threads"], li["plan_build_ms"], li["plan_upload_ms"]))


def test_c5_particle_filter_cycle_131072_particles(oracle, base):
    """forecast (one day) -> analysis -> forecast: one GPU's share of the 1 M-particle cycle"""
    n, T1 = 131072, 48
    clim = year_clim(n=2 * T1, start_day=150)           # two summer days
    flags = sa.flags_from()
    members = synth.perturbed_params(base, n)
    twin = build(flags, [clim], members, sa.F32_MIXED)
    twin.run(0, T1, want_planes=False)
    twin_state = twin.get_state()
    twin2, _ = twin.run(T1, T1)
    twin.close()

    b = build(flags, [clim], members, sa.F32_MIXED)
    p1, _ = b.run(0, T1)
    li = b.last_launch()
    assert li["kernel"] == "stepFastKernel<float, true, 0, 1, false>" and li["grid"] == 2048, li
    pick = np.r_[0:24, n // 2:n // 2 + 16, n - 24:n]
    want, final, so = oracle.run_block(flags, members[pick], clim.slice(0, T1))
    assert (so == 0).all()
    got = p1[:, :, torch.from_numpy(pick).to(p1.device)].double().cpu().numpy()
    print("C5 forecast, 64 sampled particles: max|d| %.3e" % np.abs(got - want).max())
    assert np.abs(got - want).max() < TOL_F32

    # analysis: weights, ancestors (oracle/pf_oracle.py)
    tot = p1[0].double().sum(0)
    obs, sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
    st = b.get_status()
    logw = b.pf_log_weights(p1[0], obs, sigma)
    np.testing.assert_allclose(logw.cpu().numpy(), po.log_weights(p1[0].cpu().numpy(), obs, sigma, st),
                               rtol=1e-12, atol=1e-12)
    anc, fixed = sd.pf_systematic_ancestors(logw, 0.5, return_fixed=True)
    fixed = fixed.cpu().numpy()
    assert np.abs(fixed - po.fixed_weights(logw.cpu().numpy())).max() <= 1
    anc_np = anc.cpu().numpy()
    np.testing.assert_array_equal(anc_np, po.systematic_ancestors(fixed, 0.5))
    uniq = int(np.unique(anc_np).size)
    assert 1 < uniq < n                                   # the filter did select
    info = sd.pf_resample(b, anc, with_params=True)
    assert info["sent"] == 0
    # resampled particle j carries its ancestor's state: bit for bit vs the unfiltered twin ...
    state = b.get_state()
    np.testing.assert_array_equal(state[:, :28], twin_state[anc_np][:, :28])
    # ... and, for the sampled ancestors, the oracle's pools after the day
    j = np.nonzero(np.isin(anc_np, pick))[0][:64]
    where = {int(c): k for k, c in enumerate(pick)}
    rows = np.array([where[int(a)] for a in anc_np[j]])
    np.testing.assert_allclose(state[j, :13], final[rows, 14:27], rtol=1e-5, atol=1e-5)
    # second forecast: particle j continues exactly like particle anc[j] of the twin
    p2, _ = b.run(T1, T1)
    assert torch.equal(p2, twin2[:, :, anc.long()])
    # a setup() after the resampling re-initialises the RESAMPLED parameter sets (the converted
    # block is the only copy): particle j then reproduces member anc[j]'s first day
    b.setup()
    p3, _ = b.run(0, T1)
    assert torch.equal(p3, p1[:, :, anc.long()])
    b.close()
    print("C5 analysis: %d unique ancestors of %d, resampled state == ancestors' (twin bit-exact, oracle pools "
          "within 1e-5)" % (uniq, n))


# ---------------------------------------------------------------------------------------------
def _scenario(base, lethal):
    """60 days from day 130 (spring: leaf-on through growing degree days falls inside, GPP is on by
    day and off by night), 150 members (three chunks, the last one ragged), events of every carbon
    type; with `lethal` also a clear-cut that kills every stand, a re-planting and a member dead
    from the start.  The launch is cut at odd steps (a one-step tile tail included)."""
    clim = year_clim(n=48 * 60, start_day=129)
    ev = []

    def add(day, typ, *p):
        e = sa.Event(); e.type = typ; e.year = int(clim.year[0]); e.day = 129 + day
        for i, v in enumerate(p):
            e.p[i] = v
        ev.append(e)
    add(4, 2, 1.5, 0)                          # canopy irrigation
    add(6, 4, 0.4)                             # tillage
    add(9, 1, 0.3, 0.2, 0.1, 0.1)              # partial harvest
    add(9, 0, 2.0, 30.0, 1.0)                  # fertiliser, same day
    if lethal:
        add(20, 1, 1.0, 1.0, 0.0, 0.0)         # clear-cut: every member dies
        add(30, 3, 40.0, 300.0, 50.0, 60.0)    # re-planting
    add(41, 2, 2.0, 1)                         # soil irrigation
    members = synth.perturbed_params(base, 150)
    if lethal:
        members[5, pi("plantWoodInit")] = 0.0  # never alive, keeps its leaves
    return clim, ev, members


KERNELS = [
    ("one_wave_f64", sa.F64, sa.KERNEL_ONE_WAVE, 0, "stepFastKernel<double, true, 0, 1, false>"),
    ("one_wave_f32", sa.F32_MIXED, sa.KERNEL_ONE_WAVE, 0, "stepFastKernel<float, true, 0, 1, false>"),
    ("coop_lds_f64", sa.F64, sa.KERNEL_COOP_LDS, 0, "stepCoopKernel<double, true, true, false>"),
    ("coop_lds_f32", sa.F32_MIXED, sa.KERNEL_COOP_LDS, 0, "stepCoopKernel<float, true, true, false>"),
    ("coop_hbm_f64", sa.F64, sa.KERNEL_COOP_HBM, 0, "stepCoopKernel<double, true, false, false>"),
    ("coop_hbm_f32", sa.F32_MIXED, sa.KERNEL_COOP_HBM, 0, "stepCoopKernel<float, true, false, false>"),
    ("coop_pair_f64", sa.F64, sa.KERNEL_COOP_PAIR, 0, "stepCoopPairKernel<double, true, false>"),
    ("coop_pair_f32", sa.F32_MIXED, sa.KERNEL_COOP_PAIR, 0, "stepCoopPairKernel<float, true, false>"),
    ("coop_quad_f64", sa.F64, sa.KERNEL_COOP_QUAD, 0, "stepCoopQuadKernel<double, true>"),
    ("coop_quad_f32", sa.F32_MIXED, sa.KERNEL_COOP_QUAD, 0, "stepCoopQuadKernel<float, true>"),
    ("runtime_flags_f64", sa.F64, sa.KERNEL_ONE_WAVE, sa.KOPT_RUNTIME_FLAGS, "stepFastKernel<double, true, 1, 1, false>"),
    ("runtime_flags_f32", sa.F32_MIXED, sa.KERNEL_ONE_WAVE, sa.KOPT_RUNTIME_FLAGS, "stepFastKernel<float, true, 1, 1, false>"),
    # the nitrogen-cycle flag set (litter pool + anaerobic + nitrogen cycle, sipnet_amd/data/allflags_forest.param)
    # (that file's dVpdExp / soilRespMoistEffect are not 2 / 1: the general-exponent builds; "_plain" sets them so)
    ("coop_ncycle_f64", sa.F64, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNKernel<double, false>"),
    ("coop_ncycle_f32", sa.F32_MIXED, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNKernel<float, false>"),
    ("coop_ncycle_f64_plain", sa.F64, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNKernel<double, true>"),
    ("coop_ncycle_f32_plain", sa.F32_MIXED, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNKernel<float, true>"),
    ("coop_ncycle_pair_f64", sa.F64, sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNPairKernel<double, false>"),
    ("coop_ncycle_pair_f32", sa.F32_MIXED, sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNPairKernel<float, false>"),
    ("coop_ncycle_pair_f64_plain", sa.F64, sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNPairKernel<double, true>"),
    # the optional-physics instantiations (run-time flags; the flag set rides in the row's name): russell_3's family
    # on the default pools' layouts, "everything" on the nitrogen-cycle ones
    ("x_russell3_lds_f64", sa.F64, sa.KERNEL_COOP_LDS, 0, "stepCoopXKernel<double, false, true, false>"),
    ("x_russell3_lds_f32", sa.F32_MIXED, sa.KERNEL_COOP_LDS, 0, "stepCoopXKernel<float, false, true, false>"),
    ("x_russell3_hbm_f64", sa.F64, sa.KERNEL_COOP_HBM, 0, "stepCoopXKernel<double, false, false, false>"),
    ("x_russell3_pair_f64", sa.F64, sa.KERNEL_COOP_PAIR, 0, "stepCoopXPairKernel<double, false, false>"),
    ("x_russell3_pair_f32", sa.F32_MIXED, sa.KERNEL_COOP_PAIR, 0, "stepCoopXPairKernel<float, false, false>"),
    ("x_russell3_lds_f64_plain", sa.F64, sa.KERNEL_COOP_LDS, 0, "stepCoopXKernel<double, true, true, false>"),
    ("x_anaerobic_sat_flood_lds_f64", sa.F64, sa.KERNEL_COOP_LDS, 0, "stepCoopXKernel<double, false, true, false>"),
    ("x_anaerobic_sat_flood_pair_f64", sa.F64, sa.KERNEL_COOP_PAIR, 0, "stepCoopXPairKernel<double, false, false>"),
    ("x_anaerobic_sat_flood_pair_f32", sa.F32_MIXED, sa.KERNEL_COOP_PAIR, 0, "stepCoopXPairKernel<float, false, false>"),
    # ... four chunks per workgroup: the fp32-mixed build only (the fp64 one would spill; the engine refuses it)
    ("x_russell3_quad_f32", sa.F32_MIXED, sa.KERNEL_COOP_QUAD, 0, "stepCoopXQuadKernel<float, false>"),
    ("x_anaerobic_sat_flood_quad_f32", sa.F32_MIXED, sa.KERNEL_COOP_QUAD, 0, "stepCoopXQuadKernel<float, false>"),
    ("x_russell3_quad_f32_plain", sa.F32_MIXED, sa.KERNEL_COOP_QUAD, 0, "stepCoopXQuadKernel<float, true>"),
    ("x_everything_ncycle_f64", sa.F64, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNXKernel<double, false>"),
    ("x_everything_ncycle_f32", sa.F32_MIXED, sa.KERNEL_COOP_NCYCLE, 0, "stepCoopNXKernel<float, false>"),
    ("x_everything_ncycle_pair_f64", sa.F64, sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNXPairKernel<double, false>"),
    ("x_everything_ncycle_pair_f64_plain", sa.F64, sa.KERNEL_COOP_NCYCLE_PAIR, 0, "stepCoopNXPairKernel<double, true>"),
]
NCYCLE_FLAGS = dict(litterPool=1, anaerobic=1, nitrogenCycle=1)
X_FLAGS = {"russell3": dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0),
           "anaerobic_sat_flood": dict(anaerobic=1, litterPool=1, carbonSaturation=1, flooding=1, growthResp=1, leafWater=1),
           "everything": dict(carbonSaturation=1, flooding=1, growthResp=1, leafWater=1, **NCYCLE_FLAGS)}


@pytest.mark.parametrize("name,prec,kernel,options,expect", KERNELS, ids=[k[0] for k in KERNELS])
def test_every_throughput_kernel_instantiation_against_the_oracle(name, prec, kernel, options, expect,
                                                                  oracle, base):
    """each forced kernel vs the ORACLE: fp64 at 1e-9 incl. the rare paths (clear-cut + death,
    re-planting, a member dead from the start, ragged chunk, split launch); fp32-mixed at 2e-6 on
    the same schedule without the lethal events (emptied pools keep ~1e-5 gC of fp32 residue,
    which the fuzz test judges by time sums instead)"""
    flags = sa.flags_from()
    if name.startswith("x_"):
        flags = sa.flags_from(**X_FLAGS[[k for k in X_FLAGS if name.startswith("x_" + k)][0]])
        base = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    elif kernel in (sa.KERNEL_COOP_NCYCLE, sa.KERNEL_COOP_NCYCLE_PAIR):
        flags = sa.flags_from(**NCYCLE_FLAGS)
        base = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    clim, ev, members = _scenario(base, lethal=prec == sa.F64)
    if name.startswith("x_"):
        # spread on what only the optional flags read (the saturation level around the soil carbon stock, so that the
        # unit clip works on both sides; partial flood drainage; a canopy that holds little water)
        rng = np.random.default_rng(5)
        n = members.shape[0]
        members[:, pi("soilCSaturation")] = members[:, pi("soilInit")] * rng.uniform(0.5, 3.0, n)
        members[:, pi("waterDrainFrac")] = rng.uniform(0.2, 1.0, n)
        members[:, pi("leafPoolDepth")] *= rng.uniform(0.02, 1.5, n)
        members[:, pi("growthRespFrac")] *= rng.uniform(0.5, 1.5, n)
        members[:, pi("litterBreakdownRate")] *= rng.uniform(0.5, 2.0, n)
    if name.endswith("_plain"):
        members[:, pi("dVpdExp")] = 2.0
        members[:, pi("soilRespMoistEffect")] = 1.0
    b = build(flags, [clim], members, prec, kernel, options, events=ev)
    T = clim.n_steps
    planes, _ = b.alloc_outputs(T)
    for a, z in ((0, 7), (7, 1000), (1000, 1015), (1015, T)):
        b.run(a, z - a, planes=planes[:, a:z])
    li = b.last_launch()
    got = planes.double().cpu().numpy()
    state = b.get_state()
    st = b.get_status()
    b.close()
    assert li["kernel"] == expect, li
    want, final, so = oracle.run_block(flags, members, clim, ev)
    assert (st == 0).all() and (so == 0).all()
    d = np.abs(got - want).max(axis=(1, 2))
    print("%s [%s]: max|dNEE| %.3e |dGPP| %.3e |dET| %.3e" % (name, li["kernel"], d[0], d[1], d[2]))
    assert d.max() < (TOL_F64 if prec == sa.F64 else TOL_F32)
    assert want[1].max() > 0.05 and (want[1] == 0).any()         # photosynthesis by day, none by night
    if prec == sa.F64:
        np.testing.assert_allclose(state[:, :13], final[:, 14:27], rtol=1e-9, atol=1e-9)
        assert state[0, 30] >= 0 and state[0, 0] > 100.0        # died at the clear-cut, wood is back


def test_non_plain_exponents_take_the_general_instantiations(oracle, base):
    """members with dVpdExp != 2 or soilRespMoistEffect != 1 need the PlainExp = false builds of
    both throughput kernels (pow through exp2 / OCML)"""
    flags = sa.flags_from()
    clim = year_clim(n=48 * 40, start_day=150)
    members = synth.perturbed_params(base, 130)
    members[3, pi("dVpdExp")] = 1.7
    members[70, pi("soilRespMoistEffect")] = 1.4
    want, _, so = oracle.run_block(flags, members, clim)
    assert (so == 0).all()
    for kernel, expect in ((sa.KERNEL_COOP_LDS, "stepCoopKernel<double, false, true, false>"),
                           (sa.KERNEL_COOP_HBM, "stepCoopKernel<double, false, false, false>"),
                           (sa.KERNEL_COOP_PAIR, "stepCoopPairKernel<double, false, false>"),
                           (sa.KERNEL_COOP_QUAD, "stepCoopQuadKernel<double, false>"),
                           (sa.KERNEL_ONE_WAVE, "stepFastKernel<double, false, 0, 1, false>")):
        b = build(flags, [clim], members, sa.F64, kernel)
        got = b.run()[0].cpu().numpy()
        li = b.last_launch()
        b.close()
        assert li["kernel"] == expect, li
        assert np.abs(got - want).max() < TOL_F64, expect


def test_regular_tile_path_equals_the_general_step_bit_for_bit(oracle, base):
    """the cooperative kernel's carbon wave runs regular 16-step tiles (uniform step length, the
    steady one- or two-eviction ring pattern, no events / year roll-over / pending phenology, every
    member of the wavefront alive) without per-step records, and everything else through the
    general step.  Which path a wavefront takes depends on its neighbours (one dead member sends 64
    members down the general path), so the two must give the same bits: the same ensemble with the
    path disabled (SIPNET_KOPT_NO_REGULAR_TILES), with a never-alive member in one chunk, over a
    whole year (spring leaf-on and autumn leaf-off fall inside tiles), split at odd steps"""
    flags = sa.flags_from()
    clim = year_clim()
    members = synth.perturbed_params(base, 192)
    members[70, pi("plantWoodInit")] = 0.0               # chunk 1 never leaves the general path
    members[130] = members[3]                            # the same member in chunk 0 and chunk 2
    T = clim.n_steps
    outs = {}
    for kernel in (sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_HBM, sa.KERNEL_COOP_PAIR, sa.KERNEL_COOP_QUAD):
        for opt in (0, sa.KOPT_NO_REGULAR_TILES):
            b = build(flags, [clim], members, sa.F64, kernel, opt)
            planes, _ = b.alloc_outputs(T)
            for a, z in ((0, 5), (5, 8003), (8003, T)):
                b.run(a, z - a, planes=planes[:, a:z])
            outs[(kernel, opt)] = (planes.cpu().numpy(), b.get_state(), b.get_rings())
            b.close()
        fastp, gen = outs[(kernel, 0)], outs[(kernel, sa.KOPT_NO_REGULAR_TILES)]
        np.testing.assert_array_equal(fastp[0], gen[0])
        np.testing.assert_array_equal(fastp[1], gen[1])
        np.testing.assert_array_equal(fastp[2], gen[2])
        assert np.array_equal(fastp[0][:, :, 130], fastp[0][:, :, 3])
    np.testing.assert_array_equal(outs[(sa.KERNEL_COOP_LDS, 0)][0], outs[(sa.KERNEL_COOP_HBM, 0)][0])
    np.testing.assert_array_equal(outs[(sa.KERNEL_COOP_LDS, 0)][0], outs[(sa.KERNEL_COOP_PAIR, 0)][0])
    np.testing.assert_array_equal(outs[(sa.KERNEL_COOP_LDS, 0)][0], outs[(sa.KERNEL_COOP_QUAD, 0)][0])
    pick = np.r_[0:8, 64:72, 128:136]
    want, _, _ = oracle.run_block(flags, members[pick], clim)
    assert np.abs(outs[(sa.KERNEL_COOP_LDS, 0)][0][:, :, pick] - want).max() < TOL_F64


@pytest.mark.parametrize("kernel,per", [(sa.KERNEL_COOP_PAIR, 2), (sa.KERNEL_COOP_QUAD, 4)], ids=["pair", "quad"])
@pytest.mark.parametrize("S,M", [(8, 192), (3, 130), (3, 200), (16, 64), (1, 64), (8, 320)])
def test_multi_chunk_workgroups_cover_every_chunk_once(S, M, kernel, per, oracle, base):
    """stepCoopPairKernel / stepCoopQuadKernel map two / four chunks to a workgroup: with the
    XCD-grouped mapping (8 | S) and without, with chunk counts that leave the last workgroup partly
    empty, ragged last chunks, a single chunk.  Same bits as the one-chunk workgroups, state and
    rings included; two members of every site against the oracle"""
    flags = sa.flags_from()
    clims = [year_clim(site=s, n=48 * 12, start_day=170) for s in range(S)]
    members = synth.perturbed_params(base, M)
    outs = {}
    for k in (sa.KERNEL_COOP_HBM, kernel):
        b = build(flags, clims, members, sa.F64, k)
        planes, _ = b.alloc_outputs(clims[0].n_steps)
        planes.fill_(float("nan"))
        for a, z in ((0, 21), (21, clims[0].n_steps)):
            b.run(a, z - a, planes=planes[:, a:z])
        li = b.last_launch()
        outs[k] = (planes.cpu().numpy(), b.get_state(), b.get_rings(), li)
        b.close()
    li = outs[kernel][3]
    chunks = S * ((M + 63) // 64)
    assert li["kernel"] == ("stepCoopPairKernel<double, true, false>" if per == 2 else "stepCoopQuadKernel<double, true>")
    assert li["block_threads"] == (512 if per == 2 else 768)
    assert li["grid"] == (8 * ((chunks // 8 + per - 1) // per) if S % 8 == 0 else (chunks + per - 1) // per), li
    for k in range(3):
        np.testing.assert_array_equal(outs[kernel][k], outs[sa.KERNEL_COOP_HBM][k])
    got = outs[kernel][0].reshape(3, -1, S, M)
    assert np.isfinite(got).all()
    for s in range(S):
        pick = np.array([0, M - 1])
        want, _, _ = oracle.run_block(flags, members[pick], clims[s])
        assert np.abs(got[:, :, s, pick] - want).max() < TOL_F64, s


def test_quad_kernel_has_no_full_state_build(base):
    """a forced four-chunk kernel refuses records / diagnostics loudly; AUTO falls back to the
    one-wave kernel's Full build for such a batch"""
    clim = year_clim(n=48)
    members = synth.perturbed_params(base, 64)
    b = build(sa.flags_from(), [clim], members, sa.F64, sa.KERNEL_COOP_QUAD)
    with pytest.raises(sa.SipnetError, match="no full-state"):
        b.run(full=True)
    b.close()


def test_a_member_starving_in_the_middle_of_a_regular_tile(oracle, base):
    """no events: one stand's root pools decay below the survival threshold at step 2 508, the
    thirteenth step of a regular 16-step tile, while its 63 neighbours live on.  The carbon wave
    leaves the regular path on that step (its tail is finished after the loop with the mortality
    code) and must stay on the general step as a whole wavefront from then on -- every layout of
    the cooperative kernel against the oracle, the starving member and its neighbours"""
    flags = sa.flags_from()
    clim = year_clim(n=48 * 70)
    members = synth.perturbed_params(base, 192)
    members[3, pi("plantWoodInit")] *= 2.65e-10
    pick = np.r_[0:8, 60:68]
    want, _, so = oracle.run_block(flags, members[pick], clim)
    assert (so == 0).all()
    _, rec, _ = oracle.run_member(flags, members[3], clim, None)
    roots = rec[:, 20] + rec[:, 21]
    died = int(np.nonzero(roots == 0.0)[0][0])
    assert 1000 < died < clim.n_steps - 200 and died % 16 not in (0, 15), died   # in the middle of a tile
    for kernel in (sa.KERNEL_COOP_LDS, sa.KERNEL_COOP_HBM, sa.KERNEL_COOP_PAIR, sa.KERNEL_COOP_QUAD, sa.KERNEL_ONE_WAVE):
        b = build(flags, [clim], members, sa.F64, kernel)
        got = b.run()[0].cpu().numpy()
        state = b.get_state()
        b.close()
        d = np.abs(got[:, :, pick] - want).max(axis=(1, 2))
        print("kernel %d: max|d| %.3e; member 3 died at step %d (oracle %d)" % (kernel, d.max(), int(state[3, 30]), died))
        assert d.max() < TOL_F64, kernel
        assert int(state[3, 30]) == died


@pytest.mark.parametrize("workload", ["c4", "c3", "c10k"])
def test_run_stats_at_the_baseline_shapes_equals_the_sums_over_the_planes(workload, base):
    """sipnet_batch_run_stats at full size (every CU busy for a whole year: the hand-overs between the
    storing wavefronts and the summing one are exercised 17 520 x 512..1 024 times): the statistics block
    of the launch equals the sums over the very planes it wrote -- c4 (two-chunk workgroups: values staged
    in LDS, summed by the light wave), c3 (four-chunk, fp32), c10k (fourth wavefront reads the tiles back
    from L2) -- and the planes are those of a plain run"""
    from bench import WORKLOADS
    wl = WORKLOADS[workload]
    S, M, T = wl["sites"], wl["members"], wl["steps"]
    prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
    clims = [year_clim(site=s) for s in range(S)]
    members = synth.perturbed_params(base, M)
    b = build(sa.flags_from(), clims, members, prec)
    planes, stats = b.run_stats(0, T)
    li = b.last_launch()
    want1 = torch.stack([planes[v].double().view(T, S, M).sum(-1) for v in range(3)])
    want2 = torch.stack([(planes[v].double().view(T, S, M) ** 2).sum(-1) for v in range(3)])
    d1 = float((stats[..., 0] - want1).abs().max() / want1.abs().max())
    d2 = float((stats[..., 1] - want2).abs().max() / want2.abs().max())
    b.setup()
    ref, _ = b.run(0, T)
    same = bool(torch.equal(ref, planes))
    b.close()
    print(workload, li["kernel"], "stats vs planes: %.2e %.2e" % (d1, d2))
    assert d1 < 1e-12 and d2 < 1e-12, li
    assert same, li


# ---- the cooperative kernels' build with bounded hand-over waits (SIPNET_KOPT_BOUNDED_WAITS) -------------------------
BOUNDED = [("coop_lds", sa.KERNEL_COOP_LDS, {}), ("coop_hbm", sa.KERNEL_COOP_HBM, {}), ("coop_pair", sa.KERNEL_COOP_PAIR, {}),
           ("coop_quad", sa.KERNEL_COOP_QUAD, {}), ("ncycle", sa.KERNEL_COOP_NCYCLE, NCYCLE_FLAGS),
           ("ncycle_pair", sa.KERNEL_COOP_NCYCLE_PAIR, NCYCLE_FLAGS), ("x_lds", sa.KERNEL_COOP_LDS, X_FLAGS["russell3"]),
           ("nx_pair", sa.KERNEL_COOP_NCYCLE_PAIR, X_FLAGS["everything"])]


@pytest.mark.parametrize("name,kernel,kw", BOUNDED, ids=[k[0] for k in BOUNDED])
def test_bounded_wait_build_gives_the_products_bits_and_reports_a_wait_that_never_ends(name, kernel, kw, base):
    """step_coop_bounded.hip = step_coop.hip compiled with a budget of polls on every hand-over wait: the same planes,
    state and rings as the product's kernels, bit for bit; and with SIPNET_KOPT_WAIT_SELFTEST (the light wave stops
    posting after 100 steps) the launch ENDS -- sipnet_batch_run answers SIPNET_ERR_INTERNAL naming a wait and a step
    instead of hanging the device"""
    flags = sa.flags_from(**kw)
    if kw:
        base = sa.read_params(os.path.join(os.path.dirname(BASE), "allflags_forest.param"), flags)[0]
    clim, ev, members = _scenario(base, lethal=True)
    outs = []
    for opt in (0, sa.KOPT_BOUNDED_WAITS):
        b = build(flags, [clim], members, sa.F64, kernel, opt, events=ev)
        planes, _ = b.alloc_outputs(clim.n_steps)
        for a, z in ((0, 7), (7, 1000), (1000, clim.n_steps)):
            b.run(a, z - a, planes=planes[:, a:z])
        outs.append((planes.cpu().numpy(), b.get_state(), b.get_rings(), b.last_launch()["kernel"]))
        b.close()
    assert outs[0][3] == outs[1][3]                       # the same instantiation, from the other translation unit
    for k in range(3):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
    b = build(flags, [clim], members, sa.F64, kernel, sa.KOPT_BOUNDED_WAITS | sa.KOPT_WAIT_SELFTEST, events=ev)
    import time
    t0 = time.time()
    with pytest.raises(sa.SipnetError) as e:
        b.run(0, 1000)
    assert e.value.code == 7 and "hand-over wait" in str(e.value) and "gave up at step" in str(e.value), str(e.value)
    assert time.time() - t0 < 60
    b.close()
    # the device is fine afterwards
    b = build(flags, [clim], members, sa.F64, kernel, 0, events=ev)
    got = b.run(0, 1000)[0].cpu().numpy()
    b.close()
    np.testing.assert_array_equal(got, outs[0][0][:, :1000])
