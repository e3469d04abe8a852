"""Particle-filter analysis step on the GPU (pf.hip; BASELINE config C5):
likelihood weights, systematic resampling, packing and the resampling gather against the
numpy oracle (oracle/pf_oracle.py), and a forecast -> analysis -> forecast cycle whose
resampled particles continue exactly like their ancestors."""
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
from tests.test_pf import weights_case

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")
DEV = "cuda"


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


@pytest.fixture(scope="module")
def clim():
    return synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(48 * 12)))


def batch_of(clim, members, prec=sa.F64):
    b = sa.Batch(sa.flags_from(), 1, members.shape[0], prec, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    return b


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED])
def test_log_weights_match_oracle(base, clim, prec):
    members = synth.perturbed_params(base, 200)
    members[7, pi("leafAllocation")] = 0.9            # invalid allocation: status 3, weight -inf
    members[7, pi("woodAllocation")] = 0.9
    b = batch_of(clim, members, prec)
    planes, _ = b.run(0, 96)
    st = b.get_status()
    assert st[7] != 0 and (np.delete(st, 7) == 0).all()
    nee = planes[0]
    obs, sigma = float(nee[:, 0].double().sum()) * 0.9, 0.05
    got = b.pf_log_weights(nee, obs, sigma).cpu().numpy()
    want = po.log_weights(nee.cpu().numpy(), obs, sigma, st)
    b.close()
    assert got[7] == -np.inf
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("kind", ["uniform", "mild", "degenerate", "one"])
@pytest.mark.parametrize("n", [1, 5, 4096, 1 << 20])
def test_systematic_ancestors_match_oracle(kind, n):
    if n < 8 and kind in ("degenerate", "one"):
        pytest.skip("needs more particles")
    lw = weights_case(n, kind)
    for u0 in (0.0, 0.41, 0.999999999):
        anc, fixed = sd.pf_systematic_ancestors(torch.from_numpy(lw).to(DEV), u0, return_fixed=True)
        fixed = fixed.cpu().numpy()
        # integer weights: device exp vs glibc exp may round a weight to the neighbouring integer
        assert np.abs(fixed - po.fixed_weights(lw)).max() <= 1
        # given the same integer weights the resampling is exact
        np.testing.assert_array_equal(anc.cpu().numpy(), po.systematic_ancestors(fixed, u0))
    with pytest.raises(sa.SipnetError):
        sd.pf_systematic_ancestors(torch.full((16,), -np.inf, dtype=torch.float64, device=DEV), 0.3)


def test_ancestors_without_host_round_trip_equal_the_synchronous_call():
    """the asynchronous entry point: same ancestors, the total weight left on the device
    (0 when no particle survives, instead of the synchronous call's error)"""
    lw = weights_case(4096, "mild")
    total = torch.full((1,), -7, dtype=torch.int64, device=DEV)
    anc_s, fixed = sd.pf_systematic_ancestors(torch.from_numpy(lw).to(DEV), 0.41, return_fixed=True)
    anc_a = sd.pf_systematic_ancestors(torch.from_numpy(lw).to(DEV), 0.41, total_out=total)
    assert torch.equal(anc_s, anc_a)
    assert int(total.item()) == int(fixed.sum().item()) > 0
    sd.pf_systematic_ancestors(torch.full((16,), -np.inf, dtype=torch.float64, device=DEV), 0.3,
                               total_out=total)
    assert int(total.item()) == 0


def _block_parts(blk, n, f32):
    """a packed block [words][n] of 8-byte words -> state [32][n], ring [250][n] (floats in an fp32-mixed
    batch: 250 rows of n floats in 125 rows of words), parameters [..][n]"""
    rw = 125 if f32 else 250
    ring = blk[32:32 + rw]
    if f32:
        ring = np.ascontiguousarray(ring).reshape(-1).view(np.float32).reshape(250, n).astype(np.float64)
    return blk[:32], ring, blk[32 + rw:]


def _make_block(rng, k, f32, with_params):
    """a foreign block of k particles with recognisable contents, as words + its three parts"""
    st = rng.normal(size=(32, k))
    ring = rng.normal(size=(250, k))
    prm = rng.normal(size=(80 if with_params else 0, k))
    if f32:
        ring = ring.astype(np.float32)
        words = np.concatenate([st.reshape(-1), ring.reshape(-1).view(np.float64), prm.reshape(-1)])
        ring = ring.astype(np.float64)
    else:
        words = np.concatenate([st.reshape(-1), ring.reshape(-1), prm.reshape(-1)])
    return words, st, ring, prm


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32ring"])
@pytest.mark.parametrize("with_params", [False, True])
def test_pack_and_resample_match_numpy(base, clim, with_params, prec):
    """(fp32-mixed batches keep the ring in fp32, in the batch and in a packed block)"""
    n = 300
    f32 = prec == sa.F32_MIXED
    members = synth.perturbed_params(base, n)
    b = batch_of(clim, members, prec)
    b.run(0, 300)                                     # fills state and ring with distinct values
    state, rings = b.get_state(), b.get_rings()       # [n][32], [n][250]
    assert np.abs(rings[:, 1:200]).min() > 0
    words = 32 + (125 if f32 else 250) + (80 if with_params else 0)
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(n, 37, replace=False)).astype(np.int32)
    blk = b.pack_members(torch.from_numpy(cols).to(DEV), with_params).cpu().numpy()
    assert blk.shape == (words, 37)
    bs, br, _ = _block_parts(blk, 37, f32)
    np.testing.assert_array_equal(bs, state[cols].T)
    np.testing.assert_array_equal(br, rings[cols].T)
    # received blocks: two "ranks" of 5 and 9 foreign particles with recognisable contents
    w1, s1, g1, p1 = _make_block(rng, 5, f32, with_params)
    w2, s2, g2, p2 = _make_block(rng, 9, f32, with_params)
    recv = torch.from_numpy(np.concatenate([w1, np.zeros(0), w2])).to(DEV)
    src = np.sort(rng.integers(0, n + 14, size=n)).astype(np.int32)
    everyone = torch.arange(n, dtype=torch.int32, device=DEV)
    prm_before = _block_parts(b.pack_members(everyone, True).cpu().numpy(), n, f32)[2]
    b.resample(torch.from_numpy(src).to(DEV), recv, [5, 0, 9], with_params)
    new_state, new_rings = b.get_state(), b.get_rings()
    pool_state = np.concatenate([state.T, s1, s2], axis=1)
    pool_ring = np.concatenate([rings.T, g1, g2], axis=1)
    np.testing.assert_array_equal(new_state.T, pool_state[:, src])
    np.testing.assert_array_equal(new_rings.T, pool_ring[:, src])
    prm_after = _block_parts(b.pack_members(everyone, True).cpu().numpy(), n, f32)[2]
    if with_params:
        pool_prm = np.concatenate([prm_before, p1, p2], axis=1)
        np.testing.assert_array_equal(prm_after, pool_prm[:, src])
    else:
        np.testing.assert_array_equal(prm_after, prm_before)
    b.close()


@pytest.mark.parametrize("prec,with_params", [(sa.F64, True), (sa.F32_MIXED, True), (sa.F64, False)])
def test_cycle_resampled_particles_continue_like_their_ancestors(base, clim, prec, with_params):
    """forecast 2 days -> analysis on the 2-day NEE sum -> forecast 2 more days.  Particle j of
    the filtered batch must equal particle ancestors[j] of an unfiltered twin, bit for bit."""
    n, T1, T2 = 512, 96, 96
    if with_params:
        members = synth.perturbed_params(base, n)
    else:
        # state-only filter: particles differ in their initial pools, not in parameters that the
        # step reads (setupModel consumes the *Init parameters, sipnet.c:1905-1940)
        rng = np.random.default_rng(2)
        members = np.tile(base, (n, 1))
        for name in ("plantWoodInit", "soilInit", "litterInit", "soilWFracInit", "laiInit"):
            members[:, pi(name)] *= np.exp(rng.normal(0, 0.1, n))
        members[:, pi("soilWFracInit")] = np.clip(members[:, pi("soilWFracInit")], 0.05, 1.0)
    twin = batch_of(clim, members, prec)
    twin.run(0, T1)
    twin2, _ = twin.run(T1, T2)
    twin2 = twin2.cpu().numpy()
    twin_state = twin.get_state()
    twin.close()

    b = batch_of(clim, members, prec)
    p1, _ = b.run(0, T1)
    obs = float(np.median(p1[0].double().sum(0).cpu().numpy()))
    sigma = float(p1[0].double().sum(0).std().cpu()) * 0.5 + 1e-12
    anc, info = sd.pf_analysis(b, p1[0], obs, sigma, u0=0.43, with_params=with_params)
    anc = anc.cpu().numpy()
    assert 1.0 < info["ess"] < n and info["sent"] == 0
    assert 1 < info["unique_ancestors"] < n                # the filter did select
    p2, _ = b.run(T1, T2)
    got = p2.cpu().numpy()
    np.testing.assert_array_equal(got, twin2[:, :, anc])
    np.testing.assert_array_equal(b.get_state()[:, :28], twin_state[anc][:, :28])
    b.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_one_call_analysis_equals_the_three_steps(base, clim, prec):
    """sipnet_batch_pf_analysis (log-weights with block maxima -> ancestors -> resample in one library call) against
    the three entry points called one by one, on twin batches: same log-weights, ancestors, total weight and
    resampled state, ring and parameters, bit for bit"""
    n = 1000                                             # (a ragged last block of 256)
    members = synth.perturbed_params(base, n)
    twins = [batch_of(clim, members, prec) for _ in range(2)]
    planes = [b.run(0, 96)[0] for b in twins]
    assert torch.equal(planes[0], planes[1])
    nee = planes[0][0]
    obs, sigma = float(nee[:, 3].double().sum()), float(nee.double().sum(0).std()) * 0.7
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    anc1, logw1 = twins[0].pf_analysis_local(nee, obs, sigma, 0.37, with_params=True, total_out=total)
    logw2 = twins[1].pf_log_weights(planes[1][0], obs, sigma)
    anc2, fixed = sd.pf_systematic_ancestors(logw2, 0.37, return_fixed=True)
    twins[1].resample(anc2, None, (), True)
    assert torch.equal(logw1, logw2) and torch.equal(anc1, anc2)
    assert int(total.item()) == int(fixed.sum().item()) > 0
    assert 1 < int(torch.unique_consecutive(anc1).numel()) < n
    np.testing.assert_array_equal(twins[0].get_state(), twins[1].get_state())
    # (ring slots no step has written yet are uninitialised memory: slot 0 and the 96 inserts)
    np.testing.assert_array_equal(twins[0].get_rings()[:, :97], twins[1].get_rings()[:, :97])
    everyone = torch.arange(n, dtype=torch.int32, device=DEV)
    w = 32 + (125 if prec == sa.F32_MIXED else 250)
    assert torch.equal(twins[0].pack_members(everyone, True)[w:], twins[1].pack_members(everyone, True)[w:])   # parameters
    for b in twins:
        b.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32ring"])
def test_two_ranks_emulated_on_one_gpu(base, clim, prec):
    """two batches stand in for two ranks: exchange plan + pack + resample with received blocks
    reproduce the global gather (the all-to-all itself is covered over gloo in test_pf.py)"""
    n, world = 256, 2
    members = synth.perturbed_params(base, n * world)
    ranks = [batch_of(clim, members[r * n:(r + 1) * n], prec) for r in range(world)]
    for b in ranks:
        b.run(0, 96)
    before = [np.concatenate([b.get_state().T, b.get_rings().T], axis=0) for b in ranks]
    lw = weights_case(n * world, "mild", seed=9)
    anc = sd.pf_systematic_ancestors(torch.from_numpy(lw).to(DEV), 0.77)
    want = po.resample_global(before, anc.cpu().numpy())
    plans = [sd.pf_exchange_plan(anc, n, world, r) for r in range(world)]
    packed = [[ranks[r].pack_members(plans[r][0][d], True) for d in range(world)] for r in range(world)]
    for r in range(world):
        _, src, counts = plans[r]
        recv = torch.cat([packed[s][r].reshape(-1) for s in range(world)])
        ranks[r].resample(src, recv, counts, True)
    for r in range(world):
        after = np.concatenate([ranks[r].get_state().T, ranks[r].get_rings().T], axis=0)
        np.testing.assert_array_equal(after, want[r])
        ranks[r].close()


@pytest.mark.parametrize("world,n,kind", [(2, 256, "mild"), (4, 1000, "mild"), (8, 4096, "mild"),
                                          (8, 4096, "degenerate"), (3, 77, "uniform"), (8, 131072, "mild")])
def test_native_exchange_plan_equals_the_torch_formulation(world, n, kind):
    """sipnet_pf_exchange_plan (kernels + one scan + one host sync) against the torch plan it
    replaces, for every rank of an emulated world: same send lists, same source map, same counts"""
    lw = weights_case(n * world, kind, seed=world)
    anc = sd.pf_systematic_ancestors(torch.from_numpy(lw).to(DEV), 0.37)
    for rank in range(world):
        got = sd.pf_exchange_plan(anc, n, world, rank)
        want = sd.pf_exchange_plan_reference(anc, n, world, rank)
        assert got[2] == want[2], (rank, got[2], want[2])
        assert torch.equal(got[1], want[1]), rank
        for d in range(world):
            assert torch.equal(got[0][d], want[0][d].to(torch.int32)), (rank, d)
    total_sent = sum(int(c.numel()) for r in range(world) for c in sd.pf_exchange_plan(anc, n, world, r)[0])
    total_recv = sum(sum(sd.pf_exchange_plan(anc, n, world, r)[2]) for r in range(world))
    assert total_sent == total_recv


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_resampled_parameter_index_equals_moving_the_rows(base, clim, prec):
    """A one-batch filter whose particles carry their parameters resamples an INDEX into the parameter bank (4 bytes
    per particle instead of 640); the one-wave kernel reads through it, every other reader gets the rows back in
    column order first.  Three cycles on the one-wave kernel (the index composed three times) against a twin whose
    cooperative kernel forces the rows to be gathered before every forecast: same planes, state, rings and -- read
    through the index / from the gathered rows -- parameters, bit for bit; then setupModel() (which reads the rows)
    and a parameter re-draw (which writes them) on the indexed batch"""
    n, T = 1000, 48
    members = synth.perturbed_params(base, n)

    def make(kernel):
        b = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True, kernel=kernel)
        b.set_climate(0, clim)
        b.set_params(0, members)
        b.setup()
        return b

    a, c = make(sa.KERNEL_ONE_WAVE), make(sa.KERNEL_COOP_HBM)
    everyone = torch.arange(n, dtype=torch.int32, device=DEV)
    w = 32 + (125 if prec == sa.F32_MIXED else 250)
    lineage = np.arange(n)
    for cyc in range(3):
        pa, _ = a.run(cyc * T, T)
        pc, _ = c.run(cyc * T, T)
        assert a.last_launch()["kernel"].startswith("stepFastKernel") and c.last_launch()["kernel"].startswith("stepCoopKernel")
        np.testing.assert_allclose(pa.cpu().numpy(), pc.cpu().numpy(), rtol=0, atol=2e-9 if prec == sa.F64 else 5e-5)
        # the SAME weights for both (the one-wave kernel's plane), so that both resample the same ancestors
        nee = pa[0]
        tot = nee.double().sum(0)
        obs, sigma = float(tot.median()), float(tot.std()) * 0.6 + 1e-12
        anc_a, _ = a.pf_analysis_local(nee, obs, sigma, 0.21 + 0.2 * cyc, with_params=True)
        anc_c, _ = c.pf_analysis_local(nee, obs, sigma, 0.21 + 0.2 * cyc, with_params=True)
        assert torch.equal(anc_a, anc_c) and 1 < int(torch.unique_consecutive(anc_a).numel()) < n
        lineage = lineage[anc_a.cpu().numpy()]
        prm_a, prm_c = a.pack_members(everyone, True)[w:], c.pack_members(everyone, True)[w:]
        assert torch.equal(prm_a, prm_c)
        # ... and they are the rows of the original draw's columns `lineage` (converted parameters: compare two columns
        # that descend from the same draw, and a particle with its ancestor's row before the resampling)
        same = np.flatnonzero(lineage == lineage[0])
        assert torch.equal(prm_a[:, same[0]], prm_a[:, same[-1]])
    assert len(np.unique(lineage)) < n
    # setupModel() on the indexed batch: every column starts again from the parameters it carries NOW
    a.setup()
    c.setup()
    np.testing.assert_array_equal(a.get_state(), c.get_state())
    pa, _ = a.run(0, T)
    pc, _ = c.run(0, T)
    np.testing.assert_allclose(pa.cpu().numpy(), pc.cpu().numpy(), rtol=0, atol=2e-9 if prec == sa.F64 else 5e-5)
    assert torch.equal(a.pack_members(everyone, True)[w:], c.pack_members(everyone, True)[w:])
    # a re-draw after an analysis: new rows for the first 100 columns go into column order, the rest keep what they carry
    for b in (a, c):
        b.pf_analysis_local(pa[0], float(pa[0].double().sum(0).median()), 1.0, 0.5, with_params=True)
        b.set_params(0, members[::-1][:100].copy())
        b.run(T, T)
    assert torch.equal(a.pack_members(everyone, True)[w:], c.pack_members(everyone, True)[w:])
    np.testing.assert_allclose(a.get_state()[:, :13], c.get_state()[:, :13], rtol=1e-6 if prec == sa.F32_MIXED else 1e-11, atol=1e-9)
    a.close()
    c.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED])
def test_log_weights_from_the_forecast_launch_equal_the_analysis_own(base, clim, prec):
    """sipnet_batch_pf_arm: the one-wave forecast kernel leaves the log-weights of its NEE sum; the analysis that follows
    skips its first phase.  Log-weights, ancestors, total weight and the resampled state are bit-identical to the
    unarmed cycle -- also with a member that did not run (-inf) and with members past the last full wavefront."""
    M = 64 * 37 + 19
    members = synth.perturbed_params(base, M, seed=11)
    members[70, pi("leafAllocation")] = 0.9            # status 3: weight -inf
    members[70, pi("woodAllocation")] = 0.9
    res = {}
    for armed in (False, True):
        b = sa.Batch(sa.flags_from(), 1, M, prec, fast_math=True, kernel=sa.KERNEL_ONE_WAVE)
        b.set_climate(0, clim)
        b.set_params(0, members)
        b.setup()
        planes, _ = b.alloc_outputs(48)
        out = []
        for day in range(3):
            obs, sigma = -0.02 * (day + 1), 0.3
            if armed:
                b.pf_arm(obs, sigma)
            b.run(day * 48, 48, planes=planes)
            tot = torch.zeros(1, dtype=torch.int64, device=planes.device)
            anc, logw = b.pf_analysis_local(planes[0], obs, sigma, 0.37, with_params=True, total_out=tot)
            out.append((anc.clone(), logw.clone(), tot.clone()))
        # (an analysis with OTHER arguments than the armed ones falls back to its own first phase)
        if armed:
            b.pf_arm(1.0, 0.3)
        b.run(3 * 48, 48, planes=planes)
        anc, logw = b.pf_analysis_local(planes[0], 2.0, 0.3, 0.11, with_params=True)
        out.append((anc.clone(), logw.clone(), None))
        res[armed] = (out, b.get_state().copy(), b.get_rings().copy())
        assert "stepFastKernel" in b.last_launch()["kernel"]
        b.close()
    for (a0, l0, t0), (a1, l1, t1) in zip(res[False][0], res[True][0]):
        assert torch.equal(a0, a1) and torch.equal(l0.view(torch.int64), l1.view(torch.int64))
        assert t0 is None or torch.equal(t0, t1)
    assert torch.isinf(res[True][0][0][1][70])
    assert np.array_equal(res[False][1], res[True][1], equal_nan=True) and np.array_equal(res[False][2], res[True][2], equal_nan=True)


# ---- round 6: the one-launch analysis' geometry, residency guard and way out -----------------------------------------
def _analysis_reference(b, nee, obs, sigma, u0):
    """the three entry points one by one (multi-launch path): ancestors, integer weights"""
    logw = b.pf_log_weights(nee, obs, sigma)
    anc, fixed = sd.pf_systematic_ancestors(logw, u0, return_fixed=True)
    return anc, fixed, logw


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_one_launch_analysis_geometry_follows_the_device_not_a_constant(base, clim, prec):
    """The grid of pfFusedKernel is what the device can keep resident (occupancy x compute units / filters sharing the
    device), not a compile-time 512: a batch that believes it has 32 compute units (one partition of a CPX-mode MI355X),
    2, or a sixth of the device, and the multi-launch path when next to nothing is left -- all with the ancestors,
    log-weights and total weight of the reference path, and the oracle's ancestors for those integer weights.  With few
    workgroups a thread owns SEVERAL consecutive slots (the serial sums + one block scan per workgroup of round 6)."""
    n = 64 * 150 + 37
    members = synth.perturbed_params(base, n, seed=5)
    members[11, pi("leafAllocation")] = 0.9            # status 3: weight -inf
    members[11, pi("woodAllocation")] = 0.9
    ref = batch_of(clim, members, prec)
    planes, _ = ref.run(0, 48)
    nee = planes[0]
    tot = nee.double().sum(0)
    obs, sigma = float(tot.median()), float(tot[torch.isfinite(tot)].std()) * 0.5 + 1e-12
    anc_ref, fixed, logw_ref = _analysis_reference(ref, nee, obs, sigma, 0.37)
    np.testing.assert_array_equal(anc_ref.cpu().numpy(), po.systematic_ancestors(fixed.cpu().numpy(), 0.37))
    ref.close()
    seen = set()
    for cus, share, opts in ((256, 1, 0), (32, 1, 0), (2, 1, 0), (256, 6, 0), (1, 6, 0), (256, 1, sa.KOPT_PF_MULTI_LAUNCH)):
        b = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True, kernel_options=opts)
        b.set_climate(0, clim)
        b.set_params(0, members)
        b.setup()
        p, _ = b.run(0, 48)
        assert torch.equal(p[0], nee)
        b.debug_set_num_cus(cus)
        b.set_device_share(share)
        total = torch.zeros(1, dtype=torch.int64, device=DEV)
        anc, logw = b.pf_analysis_local(p[0], obs, sigma, 0.37, with_params=True, total_out=total)
        info = b.pf_info()
        assert torch.equal(anc, anc_ref) and torch.equal(logw.view(torch.int64), logw_ref.view(torch.int64)), (cus, share, opts, info)
        assert int(total.item()) == int(fixed.sum().item()) > 0
        assert info["device_share"] == share
        if info["fused"] == 0:
            assert opts or info["budget"] < 8 or (n + 255) // 256 > 16 * info["budget"], info
            assert info["fused"] == 0 and info["grid"] == 0, info
        else:
            assert info["fused"] == 1 and 1 <= info["grid"] <= min(info["budget"], 512), info
            assert info["budget"] <= 512, info
        seen.add((info["fused"], info["grid"]))
        b.close()
    assert any(f == 1 and 0 < g * 256 < n for f, g in seen), seen        # several slots per thread did happen
    assert any(f == 0 for f, g in seen) and len(seen) >= 3, seen


def test_a_grid_that_is_not_co_resident_ends_with_an_error_not_a_hang(base, clim):
    """A workgroup of the one-launch analysis that never arrives (test hook) stands for one that was never scheduled: the
    others' polls run out of budget, the launch is void -- total weight INT64_MIN, SIPNET_ERR_INTERNAL from the
    synchronous call -- and the NEXT analysis of the same batch is not affected (round 5's counters grew across launches:
    every later launch would have spun for ever)."""
    n = 64 * 64
    members = synth.perturbed_params(base, n, seed=6)
    b = sa.Batch(sa.flags_from(), 1, n, sa.F64, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    p, _ = b.run(0, 48)
    tot = p[0].double().sum(0)
    obs, sigma = float(tot.median()), float(tot.std()) * 0.5 + 1e-12
    anc_ref, fixed, _ = _analysis_reference(b, p[0], obs, sigma, 0.37)
    state0 = b.get_state().copy()
    # (a) asynchronous: the void mark is where the caller looks for the total weight
    b.debug_pf_barrier(spin_budget=4096, absent_workgroup=3)
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    b.pf_analysis_local(p[0], obs, sigma, 0.37, with_params=True, total_out=total)
    assert int(total.item()) == sa.PF_VOID_TOTAL
    # (b) synchronous: an error that names the barrier
    b.set_state(state0)
    b.debug_pf_barrier(spin_budget=4096, absent_workgroup=0)
    with pytest.raises(sa.SipnetError) as e:
        b.pf_analysis_local(p[0], obs, sigma, 0.37, with_params=True)
    assert e.value.code == 7 and "gave up" in str(e.value), str(e.value)      # SIPNET_ERR_INTERNAL
    # (c) afterwards: as if nothing had happened, 70 times (the ring of barrier sets goes round)
    b.debug_pf_barrier(spin_budget=0, absent_workgroup=-1)
    for k in range(70):
        b.set_state(state0)
        anc, _ = b.pf_analysis_local(p[0], obs, sigma, 0.37, with_params=True, total_out=total)
        if k in (0, 33, 69):
            assert torch.equal(anc, anc_ref) and int(total.item()) == int(fixed.sum().item())
    b.close()


def test_random_weights_sizes_and_geometries_against_the_oracle(base, clim):
    """the one-launch analysis over crafted log-weights (a block handed to sipnet_batch_pf_resample_peers of an unconnected
    batch: world = 1) for random particle counts, weight patterns (uniform, mild, a handful of heavy particles, one survivor,
    runs of -inf, a +-30 spread), u0 and pretended device sizes: the ancestors are the multi-launch path's and the numpy
    oracle's for the device's integer weights; the total weight their sum"""
    rng = np.random.default_rng(20261003)
    tried = set()
    for trial in range(28):
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 4096, 10007, 65536, 150001]))
        kind = str(rng.choice(["uniform", "mild", "degenerate", "one", "holes", "spread"]))
        if n < 8 and kind in ("degenerate", "one", "holes"):
            kind = "mild"
        if kind in ("holes", "spread"):
            lw = -0.5 * rng.normal(size=n) ** 2 * (30.0 if kind == "spread" else 1.0)
            if kind == "holes":
                for _ in range(int(rng.integers(1, 6))):
                    a = int(rng.integers(0, n))
                    lw[a:a + int(rng.integers(1, max(2, n // 3)))] = -np.inf
                lw[int(rng.integers(0, n))] = -0.1            # (somebody survives)
        else:
            lw = weights_case(n, kind, seed=int(rng.integers(1 << 30)))
        u0 = float(rng.choice([0.0, 0.999999999, rng.random()]))
        cus = int(rng.choice([1, 2, 8, 32, 256]))
        share = int(rng.choice([1, 1, 3, 6]))
        b = sa.Batch(sa.flags_from(), 1, n, sa.F64, fast_math=True)
        b.set_climate(0, clim)
        b.set_params(0, np.tile(base, (n, 1)))
        b.setup()
        b.debug_set_num_cus(cus)
        b.set_device_share(share)
        L = b.pf_block_len()
        block = torch.full((1, L), -np.inf, dtype=torch.float64, device=DEV)
        block[0, :n] = torch.from_numpy(lw).to(DEV)
        P = L - n
        pad = torch.full((P * 256,), -np.inf, dtype=torch.float64, device=DEV)
        pad[:n] = block[0, :n]
        block[0, n:] = pad.view(P, 256).max(dim=1).values
        total = torch.zeros(1, dtype=torch.int64, device=DEV)
        anc = b.pf_resample_peers(block, u0, total_out=total).cpu().numpy()
        info = b.pf_info()
        b.close()
        ref, fixed = sd.pf_systematic_ancestors(block[0, :n].contiguous(), u0, return_fixed=True)
        fixed = fixed.cpu().numpy()
        assert np.abs(fixed - po.fixed_weights(lw)).max() <= 1, (trial, n, kind)
        np.testing.assert_array_equal(anc, ref.cpu().numpy(), err_msg=str((trial, n, kind, u0, cus, share, info)))
        np.testing.assert_array_equal(anc, po.systematic_ancestors(fixed, u0), err_msg=str((trial, n, kind, u0, cus, share, info)))
        assert int(total.item()) == int(fixed.sum()) > 0
        tried.add((info["fused"], kind))
    assert {f for f, _ in tried} == {0, 1} and len({k for _, k in tried}) >= 5, tried
