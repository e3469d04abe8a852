"""The C host's multi-GPU object (sipnet_node_*, csrc/node.cpp) beyond one member-sharded device:
whole-site shards (SURVEY 8(e) "Partitioning": BASELINE config 4's cut), ragged member shards, and the
particle filter's exchange step by peer reads (config 5) -- each against ONE batch holding everything.
A one-GPU box runs several shards on device 0 (sipnet_node_create_sharded allows a device to be listed
more than once: the all-gathers are then event-ordered device copies); with devices = [0] the same calls
go through a one-rank RCCL communicator.  Everything else -- sharding, uploads, kernels, peer tables,
gathered layouts -- is the code an 8-GPU node executes."""
import os

import numpy as np
import pytest
import torch

import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd._lib import SHARD_MEMBERS, SHARD_SITES
from sipnet_amd.node import Node
from tests import helpers

pytestmark = pytest.mark.gpu
BASE = os.path.join(helpers.REPO, "sipnet_amd", "data", "base_forest.param")
DEV = "cuda"


@pytest.fixture(scope="module")
def base():
    return sa.read_params(BASE, sa.flags_from())[0]


def site_clims(n_sites, T):
    return [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(n_sites)]


def one_batch(flags, clims, members, prec=sa.F64, per_site_members=None):
    S, M = len(clims), members.shape[0]
    b = sa.Batch(flags, S, M, prec, fast_math=True)
    for s in range(S):
        b.set_climate(s, clims[s])
        b.set_params(s, members if per_site_members is None else per_site_members[s])
    b.setup()
    return b


@pytest.mark.parametrize("devices,n_sites", [([0, 0], 6), ([0, 0, 0], 5), ([0], 3)], ids=["2x3", "ragged-1-2-2", "rccl-one-rank"])
def test_site_sharded_node_equals_one_batch_of_all_sites(base, devices, n_sites):
    """shard k owns sites [k S / N, (k + 1) S / N) with all members: planes, statistics (concatenated along the
    site axis by gather_stats) and member status equal ONE batch of all sites bit for bit; each shard holds
    only its own sites' plans; a different member matrix at one site reaches that site's owner only"""
    M, T = 130, 48 * 6
    flags = sa.flags_from()
    clims = site_clims(n_sites, T)
    members = synth.perturbed_params(base, M)
    odd = synth.perturbed_params(base, M, seed=77)           # site 1 gets another draw
    per_site = [odd if s == 1 else members for s in range(n_sites)]
    b = one_batch(flags, clims, members, per_site_members=per_site)
    planes, stats = b.run_stats(0, T)
    want_planes = planes.cpu().numpy().reshape(3, T, n_sites, M)
    want_stats = stats.cpu().numpy()
    b.close()

    nd = Node(flags, n_sites, M, devices=devices, shard=SHARD_SITES, fast_math=True)
    assert nd.ld == max(nd.site_range(k)[1] for k in range(nd.n)) * M
    assert [nd.site_range(k)[0] for k in range(nd.n)] == [n_sites * k // nd.n for k in range(nd.n)]
    assert sum(nd.site_range(k)[1] for k in range(nd.n)) == n_sites
    if len(devices) > 1:
        assert "event-ordered" in nd.collective_library()
    else:
        assert "RCCL" in nd.collective_library()
    for s in range(n_sites):
        nd.set_climate(s, clims[s])
    nd.set_params(None, members)                                # SIPNET_ALL_SITES: one upload per shard
    nd.set_params(1, odd)
    nd.setup()
    nd.run(0, T)
    got_stats = nd.gather_stats()
    got_planes = nd.member_planes()
    np.testing.assert_array_equal(got_planes, want_planes)
    np.testing.assert_array_equal(got_stats, want_stats)
    assert (nd.status() == 0).all()
    # every shard's copy of the gathered statistics is the same, and holds shard k's sites in block k
    g0, gl = nd.gathered_stats(0), nd.gathered_stats(nd.n - 1)
    np.testing.assert_array_equal(g0, gl)
    for k in range(nd.n):
        s0, sc = nd.site_range(k)
        np.testing.assert_array_equal(g0[k][:, :, :sc], want_stats[:, :, s0:s0 + sc])
        assert (g0[k][:, :, sc:] == 0).all()
        # plan memory: a shard's batch was created with its own sites only
        assert nd.L.sipnet_batch_ncol(nd.L.sipnet_node_batch(nd.h, k)) == sc * M
    # a shorter second run (any length may be gathered), member-resolved planes gathered
    nd.setup()
    nd.run(0, T // 2)
    nd.gather_planes()
    nd.sync()
    gp = nd.gathered_planes(nd.n - 1)
    for k in range(nd.n):
        s0, sc = nd.site_range(k)
        np.testing.assert_array_equal(gp[k][:, :, :sc * M].reshape(3, T // 2, sc, M), want_planes[:, :T // 2, s0:s0 + sc])
        assert (gp[k][:, :, sc * M:] == 0).all()
    nd.close()


@pytest.mark.parametrize("devices,shard,n_sites,nseg", [([0, 0], SHARD_MEMBERS, 2, 5), ([0, 0, 0], SHARD_SITES, 5, 3),
                                                         ([0], SHARD_MEMBERS, 1, 4)],
                         ids=["members-2-shards", "sites-ragged-3-shards", "rccl-one-rank"])
def test_run_gathering_overlapped_plane_gather_equals_one_batch(base, devices, shard, n_sites, nseg):
    """sipnet_node_run_gathering: the run in segments, segment j's planes all-gathered on the shards' second streams
    under segment j + 1 -- what EVERY device holds afterwards equals ONE batch's planes bit for bit (first from a
    non-zero step0 cut at odd places, then a second, shorter call over the same buffers), and a plain run afterwards
    still gathers statistics"""
    M, T = 150, 48 * 7
    flags = sa.flags_from()
    clims = site_clims(n_sites, T)
    members = synth.perturbed_params(base, M)
    b = one_batch(flags, clims, members)
    planes, stats = b.run_stats(0, T)
    want = planes.cpu().numpy().reshape(3, T, n_sites, M)
    want_stats = stats.cpu().numpy()
    b.close()

    nd = Node(flags, n_sites, M, devices=devices, shard=shard, fast_math=True)
    for s in range(n_sites):
        nd.set_climate(s, clims[s])
    nd.set_params(None, members)
    nd.setup()
    nd.run_gathering(0, T, nseg)
    assert nd.L.sipnet_node_n_segments(nd.h) == nseg
    firsts = [nd.gathered_segment(0, j)[0] for j in range(nseg)]
    assert firsts[0] == 0 and all(f % 16 == 0 for f in firsts) and firsts == sorted(set(firsts))
    for k in range(nd.n):
        np.testing.assert_array_equal(nd.gathered_member_planes(k), want)
    with pytest.raises(sa.SipnetError):
        nd.gather_stats()                          # a gathering run leaves no statistics
    # a second call from the state the first one left (record 0 again after setup), 7 segments of a shorter run,
    # then the rest of the year from step0 = 100 (segments cut at the plan's 16-step tiles: 112, 128, ...)
    nd.setup()
    nd.run_gathering(0, 100, 7)
    np.testing.assert_array_equal(nd.gathered_member_planes(nd.n - 1), want[:, :100])
    nd.run_gathering(100, T - 100, 4)
    assert nd.gathered_segment(0, 1)[0] % 16 == 0
    np.testing.assert_array_equal(nd.gathered_member_planes(0), want[:, 100:])
    assert (nd.status() == 0).all()
    nd.setup()
    nd.run(0, T)
    got = nd.gather_stats()
    if shard == SHARD_SITES or nd.n == 1:
        np.testing.assert_array_equal(got, want_stats)
    else:
        np.testing.assert_allclose(got, want_stats, rtol=1e-12, atol=1e-12)
    assert nd.L.sipnet_node_n_segments(nd.h) == 0
    nd.close()


@pytest.mark.parametrize("devices,shard,n_sites,prec,kernel", [
    ([0, 0], SHARD_MEMBERS, 2, sa.F64, None), ([0, 0, 0], SHARD_SITES, 5, sa.F64, None), ([0], SHARD_MEMBERS, 1, sa.F64, None),
    ([0, 0], SHARD_MEMBERS, 1, sa.F32_MIXED, None), ([0, 0], SHARD_MEMBERS, 2, sa.F32_MIXED, sa.KERNEL_ONE_WAVE), ([0, 0], SHARD_MEMBERS, 1, sa.F64, "strict")],
    ids=["members-2-shards", "sites-ragged-3-shards", "rccl-one-rank", "f32mixed-2-shards", "f32mixed-one-wave", "f64-strict"])
def test_reduced_member_resolved_gather_equals_one_batch_and_the_oracle(base, devices, shard, n_sites, prec, kernel):
    """sipnet_node_run_gathering_reduced: every member's DAILY sums of NEE / GPP / ET (groups of 48 half-hourly steps, the
    last one shorter, summed in step order on the shards' second streams under the next segment's kernel, then all-gathered:
    1 / 48 of the planes' bytes) against ONE batch's planes summed on the host in the same order -- bit for bit -- and
    against the CPU oracle's daily sums (1e-9); and the planes as floats (half the bytes) against ONE batch's planes
    rounded to float.  What EVERY device holds afterwards."""
    M, T, K = 130, 48 * 6 + 20, 48
    flags = sa.flags_from()
    clims = site_clims(n_sites, T)
    members = synth.perturbed_params(base, M)
    strict = kernel == "strict"            # (SIPNET_MATH_STRICT: the strict-order kernel has no sums build -- its planes are summed on the second stream)
    kernel = None if strict else kernel
    b = one_batch(flags, clims, members, prec)
    if strict:
        b.set_math(False)
    if kernel is not None:
        b.set_kernel(kernel)
    planes, _ = b.run(0, T)
    want = planes.cpu().numpy().reshape(3, T, n_sites, M)
    b.close()
    groups = (T + K - 1) // K
    want_sums = np.zeros((3, groups, n_sites, M))
    for t in range(T):                                  # in step order, like the kernel
        want_sums[:, t // K] += want[:, t].astype(np.float64)
    nd = Node(flags, n_sites, M, precision=prec, devices=devices, shard=shard, fast_math=(not strict) if prec == sa.F64 else None, kernel=kernel)
    for s in range(n_sites):
        nd.set_climate(s, clims[s])
    nd.set_params(None, members)
    nd.setup()
    nd.run_gathering_reduced(0, T, 3, "sums", K)
    assert nd.L.sipnet_node_n_segments(nd.h) == 3
    # shards on throughput kernels sum inside the step kernel's launch (no planes written); the strict-order kernel's planes
    # are summed on the second stream
    assert nd.L.sipnet_node_reduced_in_kernel(nd.h) == (0 if strict else 1)
    assert ("Sums" in nd.kernel_name(0)) == (not strict), nd.kernel_name(0)
    for k in range(nd.n):
        got = nd.gathered_reduced_member_rows(k)
        assert got.dtype == np.float64 and got.shape == want_sums.shape
        np.testing.assert_array_equal(got, want_sums)
    if prec == sa.F64:
        ora = helpers.load_oracle()
        ref, _, st = ora.run_block(flags, members[:3], clims[0])
        assert (st == 0).all()
        ref_sums = np.stack([ref[:, g * K:(g + 1) * K].sum(axis=1) for g in range(groups)], axis=1)    # [3][groups][3 members]
        np.testing.assert_allclose(got[:, :, 0, :3], ref_sums, rtol=0, atol=1e-9)
        # ... and the planes as floats
        nd.setup()
        nd.run_gathering_reduced(0, T, 4, "f32")
        for k in range(nd.n):
            got = nd.gathered_reduced_member_rows(k)
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, want.astype(np.float32))
    else:
        with pytest.raises(sa.SipnetError):
            nd.run_gathering_reduced(0, T, 4, "f32")    # the planes of an fp32-mixed node are floats already
    with pytest.raises(sa.SipnetError):
        nd.run_gathering_reduced(0, T, 100, "sums", K)  # more segments than groups
    # a plain run afterwards is not disturbed
    nd.setup()
    nd.run(0, T)
    assert np.isfinite(nd.gather_stats()).all()
    nd.close()


def test_site_shards_with_sites_of_different_lengths(base):
    """five sites whose forcings end at different records on three site shards (1 + 2 + 2 sites; the shards' longest
    sites differ too): planes, statistics and the overlapped plane gather equal ONE batch of the five sites bit for
    bit, and the rows past a site's end are zero in everything the node owns"""
    M, T = 70, 48 * 6
    lengths = [T // 2, T, 100, 37, T - 16]
    flags = sa.flags_from()
    clims = []
    for s_, tl in enumerate(lengths):
        raw = synth.half_hourly_year_raw(T, site=s_)
        clims.append(synth.convert_raw(synth.round_like_file({k: v[:tl] for k, v in raw.items()})))
    members = synth.perturbed_params(base, M)
    b = one_batch(flags, clims, members)
    assert b.n_steps == T
    planes = torch.zeros((3, T, 5 * M), dtype=torch.float64, device=DEV)
    _, stats = b.run_stats(0, T, planes=planes)
    want = planes.cpu().numpy().reshape(3, T, 5, M)
    want_stats = stats.cpu().numpy()
    b.close()
    for s_, tl in enumerate(lengths):
        assert (want[:, tl:, s_] == 0).all() and (want_stats[:, tl:, s_] == 0).all() and np.abs(want[0, :tl, s_]).max() > 0

    nd = Node(flags, 5, M, devices=[0, 0, 0], shard=SHARD_SITES, fast_math=True)
    for s_ in range(5):
        nd.set_climate(s_, clims[s_])
    nd.set_params(None, members)
    nd.setup()
    nd.run(0, T)
    np.testing.assert_array_equal(nd.gather_stats(), want_stats)
    np.testing.assert_array_equal(nd.member_planes(), want)
    nd.setup()
    nd.run_gathering(0, T, 4)                      # cuts at 64, 144, 208: inside, at and past the sites' ends
    for k in range(nd.n):
        np.testing.assert_array_equal(nd.gathered_member_planes(k), want)
    assert (nd.status() == 0).all()
    nd.close()


@pytest.mark.timeout(180)
def test_a_failing_shard_does_not_leave_the_others_at_the_barrier(base):
    """two site shards on one device (event-ordered all-gathers: the shards' threads meet at a host barrier per
    segment); a new forcing for a site of shard 0 without a new setup makes THAT shard's launch fail -- the call
    returns its error (the other shard gives up at the barrier instead of waiting for ever), and after setup the same
    node runs again"""
    M, T = 64, 48 * 3
    flags = sa.flags_from()
    clims = site_clims(4, T)
    members = synth.perturbed_params(base, M)
    nd = Node(flags, 4, M, devices=[0, 0], shard=SHARD_SITES, fast_math=True)
    for s in range(4):
        nd.set_climate(s, clims[s])
    nd.set_params(None, members)
    nd.setup()
    nd.run_gathering(0, T, 3)
    want = nd.gathered_member_planes(1)
    nd.set_climate(0, clims[0])                     # shard 0's plan is stale now, shard 1's is not
    with pytest.raises(sa.SipnetError) as e:
        nd.run_gathering(0, T, 3)
    assert "shard 0" in str(e.value) and "sipnet_batch_setup" in str(e.value), str(e.value)
    nd.setup()
    nd.run_gathering(0, T, 3)
    np.testing.assert_array_equal(nd.gathered_member_planes(0), want)
    nd.close()


def test_member_sharded_node_with_ragged_shards_equals_one_batch(base):
    """SIPNET_SHARD_MEMBERS over three shards, 200 members (67 / 66 / 67) at two sites: the summed statistics equal
    one batch's up to the order of the additions, the planes bit for bit; column layout site * count_k + member"""
    S, M, T = 2, 200, 48 * 5
    flags = sa.flags_from()
    clims = site_clims(S, T)
    members = synth.perturbed_params(base, M)
    b = one_batch(flags, clims, members)
    planes, stats = b.run_stats(0, T)
    want_planes = planes.cpu().numpy().reshape(3, T, S, M)
    want_stats = stats.cpu().numpy()
    b.close()
    nd = Node(flags, S, M, devices=[0, 0, 0], shard=SHARD_MEMBERS, fast_math=True)
    counts = [nd.member_range(k)[1] for k in range(3)]
    assert sorted(counts) == [66, 67, 67] and nd.ld == S * 68
    for s in range(S):
        nd.set_climate(s, clims[s])
        nd.set_params(s, members)
    nd.setup()
    nd.run(0, T)
    tot = nd.gather_stats()
    np.testing.assert_array_equal(nd.member_planes(), want_planes)
    np.testing.assert_allclose(tot, want_stats, rtol=1e-12, atol=1e-12)
    nd.gather_planes()
    nd.sync()
    gp = nd.gathered_planes(0)
    for k in range(3):
        m0, mc = nd.member_range(k)
        np.testing.assert_array_equal(gp[k][:, :, :S * mc].reshape(3, T, S, mc), want_planes[:, :, :, m0:m0 + mc])
    nd.close()


# ---- the particle filter's exchange step by peer reads -----------------------------------------------------
def _filter_twin(base, clim, members, prec, T, obs_sigma=None, u0=0.43):
    """ONE batch with every particle: forecast, analysis in one library call; -> state, rings, parameters after
    the analysis + the ancestors + (obs, sigma) used"""
    n = members.shape[0]
    b = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    p, _ = b.run(0, T)
    if obs_sigma is None:
        tot = p[0].double().sum(0)
        obs_sigma = (float(tot.median()), float(tot.std()) * 0.5 + 1e-12)
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    anc, _ = b.pf_analysis_local(p[0], obs_sigma[0], obs_sigma[1], u0, with_params=True, total_out=total)
    anc = anc.cpu().numpy().copy()
    everyone = torch.arange(n, dtype=torch.int32, device=DEV)
    w = 32 + (125 if prec == sa.F32_MIXED else 250)
    prm = b.pack_members(everyone, True)[w:].cpu().numpy()
    state = b.get_state()                      # right after the analysis
    p2, _ = b.run(T, T)                        # the resampled particles continue
    out = dict(state=state, rings=b.get_rings(), prm=prm, anc=anc, obs_sigma=obs_sigma, total=int(total.item()),
               next=p2.cpu().numpy())
    b.close()
    return out


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32ring"])
@pytest.mark.parametrize("devices,n", [([0, 0], 1024), ([0, 0, 0], 1000), ([0], 700), ([0] * 6, 6 * 64 * 9 + 5)],
                         ids=["two-shards", "ragged-three", "rccl-one-rank", "six-shards-one-device"])
def test_node_filter_cycle_equals_the_one_batch_analysis(base, devices, n, prec):
    """forecast -> sipnet_node_pf_analysis (log-weight blocks, ONE all-gather, peer-read resampling) over member
    shards on one GPU against sipnet_batch_pf_analysis over all particles in one batch: same ancestors, state,
    rings (the slots written so far) and parameters bit for bit, and the resampled particles' next forecast equal"""
    T = 96
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(2 * T)))
    members = synth.perturbed_params(base, n)
    twin = _filter_twin(base, clim, members, prec, T)
    assert 1 < len(np.unique(twin["anc"])) < n

    nd = Node(sa.flags_from(), 1, n, precision=prec, devices=devices, shard=SHARD_MEMBERS, fast_math=True)
    nd.set_climate(0, clim)
    nd.set_params(0, members)
    nd.setup()
    nd.pf_connect(with_params=True)
    nd.forecast(0, T)
    nd.pf_analysis(0, twin["obs_sigma"][0], twin["obs_sigma"][1], 0.43)
    assert nd.pf_check() == 1
    nmax = max(nd.member_range(k)[1] for k in range(nd.n))
    firsts = [nd.member_range(k)[0] for k in range(nd.n)]
    crossed = 0
    for k in range(nd.n):
        m0, mc = nd.member_range(k)
        slots = nd.pf_ancestors(k)
        glob = np.array([firsts[s // nmax] + s % nmax for s in slots])       # slot -> global particle
        np.testing.assert_array_equal(glob, twin["anc"][m0:m0 + mc])
        crossed += int(((slots // nmax) != k).sum())
        np.testing.assert_array_equal(nd.shard_state(k), twin["state"][m0:m0 + mc])
    if nd.n > 1:
        assert crossed > 0                                                   # particles did cross between the shards
    # what the exchange reports about itself: the crossing particles counted by the gather, the parameters replicated on
    # every shard (an index travels), and -- six filters analysing on ONE device at the same time -- each one's spinning grid
    # sized to its share of the resident workgroups (sipnet_batch_set_device_share, set by the node)
    infos = [nd.pf_info(k) for k in range(nd.n)]
    assert sum(i["crossing"] for i in infos) == crossed and all(i["cycles"] == 1 and i["world"] == nd.n for i in infos)
    assert all(i["params_by_index"] == 1 and i["device_share"] == len(devices) for i in infos)
    assert all(i["fused"] == 1 and i["grid"] <= i["budget"] <= 512 for i in infos), infos
    if len(devices) == 6:
        assert all(i["budget"] < 512 for i in infos), infos                  # (8 resident workgroups per CU x 256 CUs / 6 = 341)
    nd.forecast(T, T)
    got = nd.member_planes()[:, :, 0, :]
    np.testing.assert_array_equal(got, twin["next"])
    for k in range(nd.n):
        m0, mc = nd.member_range(k)
        np.testing.assert_array_equal(nd.shard_rings(k)[:, :2 * T + 1], twin["rings"][m0:m0 + mc][:, :2 * T + 1])
    nd.close()


def test_node_filter_several_cycles_and_a_dead_filter(base):
    """three cycles without a host synchronisation in between (buffers alternate, peers read the current ones),
    then an observation no particle is near: every weight underflows to zero, pf_check reports the cycle"""
    T, n = 48, 768
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(4 * T)))
    members = synth.perturbed_params(base, n)
    # the twin: one batch, three one-call analyses
    b = sa.Batch(sa.flags_from(), 1, n, sa.F32_MIXED)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    obs = []
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    for c in range(3):
        p, _ = b.run(c * T, T)
        tot = p[0].double().sum(0)
        obs.append((float(tot.median()), float(tot.std()) * 0.7 + 1e-12))
        b.pf_analysis_local(p[0], obs[-1][0], obs[-1][1], 0.1 + 0.3 * c, with_params=True, total_out=total)
    want = b.get_state()
    b.close()
    nd = Node(sa.flags_from(), 1, n, precision=sa.F32_MIXED, devices=[0, 0, 0], shard=SHARD_MEMBERS)
    nd.set_climate(0, clim)
    nd.set_params(0, members)
    nd.setup()
    nd.pf_connect(with_params=True)
    for c in range(3):
        nd.pf_arm(obs[c][0], obs[c][1])          # (a hint: these small shards take a cooperative kernel, which ignores it)
        nd.forecast(c * T, T)
        nd.pf_analysis(0, obs[c][0], obs[c][1], 0.1 + 0.3 * c)
    assert nd.pf_check() == 3
    got = np.concatenate([nd.shard_state(k) for k in range(3)])
    np.testing.assert_array_equal(got, want)
    nd.close()
    # a filter none of whose particles can run (invalid allocation: status 3, weight -inf)
    from sipnet_amd.config import param_index as pi
    bad = members[:256].copy()
    bad[:, pi("leafAllocation")] = 0.9
    bad[:, pi("woodAllocation")] = 0.9
    nd = Node(sa.flags_from(), 1, 256, precision=sa.F32_MIXED, devices=[0, 0], shard=SHARD_MEMBERS)
    nd.set_climate(0, clim)
    nd.set_params(0, bad)
    nd.setup()
    nd.pf_connect(with_params=True)
    nd.forecast(0, T)
    nd.pf_analysis(0, 0.0, 1.0, 0.5)
    with pytest.raises(sa.SipnetError) as e:
        nd.pf_check()
    assert e.value.code == 3 and "zero weight" in str(e.value)
    nd.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_the_replicated_parameter_bank_equals_moving_the_rows(base, prec):
    """Particles that carry their parameters across ranks: by default every shard holds all shards' converted parameters
    (sipnet_batch_pf_connect copies them once) and a resampled particle brings a 4-byte column number; with
    SIPNET_KOPT_PF_MOVE_PARAMS its 640 bytes of rows travel instead (round 5).  Three cycles on the ONE-WAVE kernel, which
    reads the bank through the index (the index composed three times across shards), against the rows: planes, state,
    rings bit for bit; then setupModel() on the indexed shards (the rows come back out of the bank); and new parameters on
    a connected shard must be refused at the next exchange, not silently read from the stale copies on the peers."""
    T, n = 48, 64 * 30 + 11
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(5 * T)))
    members = synth.perturbed_params(base, n, seed=31)
    out = {}
    for opt in (0, sa.KOPT_PF_MOVE_PARAMS):
        nd = Node(sa.flags_from(), 1, n, precision=prec, devices=[0, 0, 0], shard=SHARD_MEMBERS, fast_math=True,
                  kernel=sa.KERNEL_ONE_WAVE, kernel_options=opt)
        nd.set_climate(0, clim)
        nd.set_params(0, members)
        nd.setup()
        nd.pf_connect(with_params=True)
        planes = []
        for c in range(3):
            nd.forecast(c * T, T)
            planes.append(nd.member_planes()[:, :, 0, :].copy())
            tot = planes[-1][0].sum(0)
            nd.pf_analysis(0, float(np.median(tot)), float(tot.std()) * 0.6 + 1e-12, 0.15 + 0.3 * c)
        assert nd.pf_check() == 3
        assert "stepFastKernel" in nd.kernel_name(0)
        infos = [nd.pf_info(k) for k in range(3)]
        assert all(i["params_by_index"] == (0 if opt else 1) and i["cycles"] == 3 for i in infos)
        nd.forecast(3 * T, T)                                    # the thrice-resampled particles, through the index
        planes.append(nd.member_planes()[:, :, 0, :].copy())
        state = np.concatenate([nd.shard_state(k) for k in range(3)])
        rings = np.concatenate([nd.shard_rings(k)[:, :4 * T + 1] for k in range(3)])
        nd.setup()                                               # setupModel(): every column from the parameters it carries now
        nd.forecast(0, T)
        planes.append(nd.member_planes()[:, :, 0, :].copy())
        out[opt] = (planes, state, rings, sum(i["crossing"] for i in infos))
        if not opt:
            nd.set_params(0, members[:100][::-1].copy())         # ... and a re-draw on a connected filter
            nd.forecast(T, T)
            with pytest.raises(sa.SipnetError) as e:
                nd.pf_analysis(0, 0.0, 1.0, 0.5)
            assert "connect again" in str(e.value)
            nd.pf_connect(with_params=True)                      # publish + connect: a new bank
            nd.forecast(T, T)
            nd.pf_analysis(0, float(np.median(tot)), 1.0, 0.5)
            assert nd.pf_check() == 1
        nd.close()
    for a, c in zip(out[0][0], out[sa.KOPT_PF_MOVE_PARAMS][0]):
        assert np.array_equal(a, c, equal_nan=True)
    assert np.array_equal(out[0][1], out[sa.KOPT_PF_MOVE_PARAMS][1], equal_nan=True)
    assert np.array_equal(out[0][2], out[sa.KOPT_PF_MOVE_PARAMS][2], equal_nan=True)
    assert out[0][3] == out[sa.KOPT_PF_MOVE_PARAMS][3] > 0


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED], ids=["f64", "f32"])
def test_a_pretended_world_of_eight_on_one_batch(base, prec):
    """The slot count of an 8-rank filter on ONE GPU: a batch connected to a world of eight whose every member is itself
    (its descriptor eight times), the gathered buffer its own block eight times -- what bench.py --pretend-world times.
    Rank 5's particles: ancestors = the oracle's over all 8 n slots (cut to rank 5's range), the resampled state = the
    batch's own columns anc % nmax, the parameters read through the index into the eight-fold bank."""
    from oracle import pf_oracle as po
    T, n, world, rank = 48, 64 * 20 + 9, 8, 5
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(2 * T)))
    members = synth.perturbed_params(base, n, seed=41)
    b = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True, kernel=sa.KERNEL_ONE_WAVE)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    p, _ = b.run(0, T)
    state0 = b.get_state().copy()
    tot = p[0].double().sum(0)
    obs, sigma = float(tot.median()), float(tot.std()) * 0.5 + 1e-12
    d = b.pf_publish(with_params=True)
    b.pf_connect([d] * world, rank)
    L = b.pf_block_len()
    gathered = torch.empty((world, L), dtype=torch.float64, device=DEV)
    b.pf_local_weights(p[0], obs, sigma, gathered[rank])
    # (the same block everywhere would give every rank exactly its own slots back: the pretended peers' weights are shifted,
    # block maxima included, so that ranks differ in total weight and particles cross)
    mine = gathered[rank].clone()
    for r in range(world):
        gathered[r] = mine + 0.35 * (r - 3)
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    anc = b.pf_resample_peers(gathered, 0.29, total_out=total).cpu().numpy()
    info = b.pf_info()
    assert info["world"] == world and info["n_slots"] == world * n and info["params_by_index"] == 1 and info["fused"] == 1
    # (device exp vs glibc exp may round a weight to the neighbouring integer: the oracle resamples the device's own integers)
    lw_all = gathered[:, :n].reshape(-1).contiguous()
    _, fixed = sd_fixed(lw_all)
    assert np.abs(fixed - po.fixed_weights(lw_all.cpu().numpy())).max() <= 1
    want = po.systematic_ancestors(fixed, 0.29)[rank * n:(rank + 1) * n]
    np.testing.assert_array_equal(anc, want)
    assert int(total.item()) == int(fixed.sum())
    assert info["crossing"] == int((anc // n != rank).sum()) > 0
    np.testing.assert_array_equal(b.get_state(), state0[anc % n])
    # the next forecast reads every particle's parameters through the index into the eight-fold bank: a twin that resamples
    # the same columns inside ONE batch (sipnet_batch_resample, its own index) must continue bit for bit alike
    twin = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True, kernel=sa.KERNEL_ONE_WAVE)
    twin.set_climate(0, clim)
    twin.set_params(0, members)
    twin.setup()
    twin.run(0, T)
    twin.resample(torch.from_numpy((anc % n).astype(np.int32)).to(DEV), with_params=True)
    p2, _ = b.run(T, T)
    q2, _ = twin.run(T, T)
    assert torch.equal(p2.view(torch.uint8), q2.view(torch.uint8))
    np.testing.assert_array_equal(b.get_state(), twin.get_state())
    b.close()
    twin.close()


def sd_fixed(logw):
    from sipnet_amd import dist as sd
    anc, fixed = sd.pf_systematic_ancestors(logw, 0.5, return_fixed=True)
    return anc, fixed.cpu().numpy()


def test_peer_resampling_of_an_unconnected_batch_is_the_one_call_analysis(base):
    """sipnet_batch_pf_local_weights + sipnet_batch_pf_resample_peers on a batch that never connected (a filter of
    one rank, no collective) = sipnet_batch_pf_analysis"""
    T, n = 96, 1000
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
    members = synth.perturbed_params(base, n)
    twin = _filter_twin(base, clim, members, sa.F64, T // 2)
    b = sa.Batch(sa.flags_from(), 1, n, sa.F64, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    p, _ = b.run(0, T // 2)
    L = b.pf_block_len()
    assert L == n + 4
    block = torch.empty((1, L), dtype=torch.float64, device=DEV)
    b.pf_local_weights(p[0], twin["obs_sigma"][0], twin["obs_sigma"][1], block[0])
    total = torch.zeros(1, dtype=torch.int64, device=DEV)
    anc = b.pf_resample_peers(block, 0.43, total_out=total)
    np.testing.assert_array_equal(anc.cpu().numpy(), twin["anc"])
    assert int(total.item()) == twin["total"] > 0
    np.testing.assert_array_equal(b.get_state(), twin["state"])
    p2, _ = b.run(T // 2, T // 2)
    np.testing.assert_array_equal(p2.cpu().numpy(), twin["next"])
    b.close()


@pytest.mark.parametrize("prec", [sa.F64, sa.F32_MIXED])
def test_the_log_weight_block_from_the_forecast_launch_equals_the_kernel_of_its_own(base, prec):
    """sipnet_batch_pf_arm with this rank's block as the target: the one-wave forecast leaves the log-weights there,
    sipnet_batch_pf_local_weights adds the 256-wide maxima -- block, ancestors, total and resampled state as unarmed"""
    T, n = 48, 64 * 41
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(2 * T)))
    members = synth.perturbed_params(base, n, seed=21)
    from sipnet_amd.config import param_index as pi
    members[100, pi("leafAllocation")] = 0.9           # status 3: -inf
    members[100, pi("woodAllocation")] = 0.9
    res = {}
    for armed in (False, True):
        b = sa.Batch(sa.flags_from(), 1, n, prec, fast_math=True, kernel=sa.KERNEL_ONE_WAVE)
        b.set_climate(0, clim)
        b.set_params(0, members)
        b.setup()
        planes, _ = b.alloc_outputs(T)
        L = b.pf_block_len()
        block = torch.full((1, L), 7.0, dtype=torch.float64, device=DEV)
        if armed:
            b.pf_arm_block(-0.03, 0.4, block[0])
        b.run(0, T, planes=planes)
        b.pf_local_weights(planes[0], -0.03, 0.4, block[0])
        total = torch.zeros(1, dtype=torch.int64, device=DEV)
        anc = b.pf_resample_peers(block, 0.61, total_out=total)
        res[armed] = (block.clone(), anc.clone(), int(total.item()), b.get_state().copy(), b.get_rings().copy())
        b.close()
    assert torch.equal(res[False][0].view(torch.int64), res[True][0].view(torch.int64))
    assert torch.equal(res[False][1], res[True][1]) and res[False][2] == res[True][2] > 0
    assert np.array_equal(res[False][3], res[True][3], equal_nan=True) and np.array_equal(res[False][4], res[True][4], equal_nan=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one-GPU boxes run the shards on device 0)")
def test_node_over_two_real_devices_equals_one_batch(base):
    """devices = [0, 1]: the thread-per-shard path, the grouped ncclAllGather over two communicators, ragged shards
    (201 members: 100 / 101), and the filter's peer reads across two GPUs -- against ONE batch on device 0"""
    S, M, T = 1, 201, 48 * 4
    flags = sa.flags_from()
    clims = site_clims(S, 2 * T)
    members = synth.perturbed_params(base, M)
    b = one_batch(flags, clims, members)
    planes, stats = b.run_stats(0, T)
    want_planes = planes.cpu().numpy().reshape(3, T, S, M)
    want_stats = stats.cpu().numpy()
    b.close()
    nd = Node(flags, S, M, devices=[0, 1], shard=SHARD_MEMBERS, fast_math=True)
    assert "RCCL" in nd.collective_library() and sorted(nd.member_range(k)[1] for k in range(2)) == [100, 101]
    nd.set_climate(0, clims[0])
    nd.set_params(0, members)
    nd.setup()
    nd.run(0, T)
    tot = nd.gather_stats()
    np.testing.assert_array_equal(nd.member_planes(), want_planes)
    np.testing.assert_allclose(tot, want_stats, rtol=1e-12, atol=1e-12)
    nd.gather_planes()
    nd.sync()
    np.testing.assert_array_equal(nd.gathered_planes(0), nd.gathered_planes(1))
    twin = _filter_twin(base, clims[0], members, sa.F64, T)
    nd.setup()
    nd.pf_connect(with_params=True)
    nd.forecast(0, T)
    nd.pf_analysis(0, twin["obs_sigma"][0], twin["obs_sigma"][1], 0.43)
    assert nd.pf_check() == 1
    for k in range(2):
        m0, mc = nd.member_range(k)
        np.testing.assert_array_equal(nd.shard_state(k), twin["state"][m0:m0 + mc])
    nd.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one-GPU boxes run the shards on device 0)")
@pytest.mark.parametrize("nseg", [1, 4])
def test_site_shards_and_the_overlapped_gather_over_two_real_devices(base, nseg):
    """devices = [0, 1], SIPNET_SHARD_SITES (config 4's cut): five sites, 3 + 2, every site's forcing, events and plan on
    ONE device; the statistics all-gather concatenated along the site axis and the member-resolved planes travelling per
    segment on the second streams (one ncclAllGather per segment and device) -- against ONE batch of all sites"""
    S, M, T = 5, 96, 48 * 3
    flags = sa.flags_from()
    clims = site_clims(S, T)
    members = synth.perturbed_params(base, M)
    b = one_batch(flags, clims, members)
    planes, stats = b.run_stats(0, T)
    want_planes = planes.cpu().numpy().reshape(3, T, S, M)
    want_stats = stats.cpu().numpy()
    b.close()
    nd = Node(flags, S, M, devices=[0, 1], shard=SHARD_SITES, fast_math=True)
    assert "RCCL" in nd.collective_library() and [nd.site_range(k)[1] for k in range(2)] in ([2, 3], [3, 2])
    for s in range(S):
        nd.set_climate(s, clims[s])
    nd.set_params(None, members)
    nd.setup()
    nd.run(0, T)
    tot = nd.gather_stats()
    np.testing.assert_array_equal(nd.member_planes(), want_planes)
    np.testing.assert_allclose(tot, want_stats, rtol=1e-12, atol=1e-12)
    nd.setup()
    nd.run_gathering(0, T, nseg)
    for k in range(2):
        np.testing.assert_array_equal(nd.gathered_member_planes(k), want_planes)
    np.testing.assert_array_equal(nd.member_planes(), want_planes)       # (a shard's own planes, walked segment by segment)
    assert (nd.status() == 0).all()
    nd.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one-GPU boxes run the shards on device 0)")
def test_filter_cycles_with_peer_reads_over_two_real_devices(base):
    """three cycles of the peer-read filter across two GPUs without a host synchronisation in between (the ONE
    all-gather per cycle is the only ordering; crossing particles are read out of the other GPU's HBM over xGMI)
    against the one-batch analysis"""
    T, n = 48, 1000
    clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(4 * T)))
    members = synth.perturbed_params(base, n)
    b = sa.Batch(sa.flags_from(), 1, n, sa.F32_MIXED)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    obs = []
    for c in range(3):
        planes, _ = b.run(c * T, T)
        tot = planes[0].double().sum(0)
        obs.append((float(tot.median()), float(tot.std()) * 1.5 + 1e-12))
        b.pf_analysis_local(planes[0], obs[-1][0], obs[-1][1], 0.1 + 0.3 * c, True, None)
    want = b.get_state()
    b.close()
    nd = Node(sa.flags_from(), 1, n, precision=sa.F32_MIXED, devices=[0, 1], shard=SHARD_MEMBERS, fast_math=True)
    nd.set_climate(0, clim)
    nd.set_params(0, members)
    nd.setup()
    nd.pf_connect(with_params=True)
    crossed = 0
    for c in range(3):
        nd.forecast(c * T, T)
        nd.pf_analysis(0, obs[c][0], obs[c][1], 0.1 + 0.3 * c)
    assert nd.pf_check() == 3
    nmax = max(nd.member_range(k)[1] for k in range(2))
    for k in range(2):
        m0, mc = nd.member_range(k)
        np.testing.assert_array_equal(nd.shard_state(k), want[m0:m0 + mc])
        crossed += int(((nd.pf_ancestors(k) // nmax) != k).sum())
    assert crossed > 0
    nd.close()
