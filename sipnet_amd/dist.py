"""Multi-GPU layout: one process per GPU, the ensemble axis sharded across ranks.

The forward model has no coupling between members, so ranks never
exchange state.  The single collective is an all-gather of the per-rank output
statistics block -- per (variable, step, site) sum and sum of squares over the rank's
members, produced by the wavefront-shuffle reduction kernel -- from which every rank
forms the ensemble mean / variance.  On ROCm the "nccl" backend is RCCL over xGMI; the
same code runs on "gloo" CPU tensors in the tests.
"""
import numpy as np


def shard_members(n_total, world, rank):
    """Contiguous member range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sites(n_sites, world, rank):
    """Contiguous site range for configurations that shard whole sites (config C4)."""
    return shard_members(n_sites, world, rank)


def _host_staged(x, group):
    """gloo has no all-gather / all-to-all on device tensors: in a rehearsal of the multi-GPU path
    on one GPU (bench.py --rehearse, tests/test_gpu_multirank.py) the exchange goes through host
    copies; on RCCL ("nccl") tensors stay on the device"""
    import torch.distributed as dist
    return x.is_cuda and dist.get_backend(group) == "gloo"


def _gather0(x, world, group):
    """all_gather_into_tensor along a new leading axis (the output is the concatenation
    along dim 0, which both RCCL and gloo accept)."""
    import torch
    import torch.distributed as dist
    x = x.contiguous()
    if _host_staged(x, group):
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype)
        dist.all_gather_into_tensor(out, x.cpu(), group=group)
        out = out.to(x.device)
    else:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
    return out.view((world,) + tuple(x.shape))


def all_gather_stats(stats, group=None):
    """stats: tensor [..., 2] (sum, sum of squares) of this rank -> [world, ..., 2].
    One all_gather_into_tensor; a single process returns stats[None]."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats.unsqueeze(0)
    world = dist.get_world_size(group)
    return _gather0(stats, world, group)


def combine_stats(gathered, counts):
    """gathered[world, ..., 2], counts[world] members per rank -> (mean, variance) of the
    whole ensemble (population variance)."""
    tot = gathered.sum(0)
    n = float(sum(counts))
    mean = tot[..., 0] / n
    var = tot[..., 1] / n - mean * mean
    return mean, var.clamp_min(0) if hasattr(var, "clamp_min") else np.maximum(var, 0)


def all_gather_planes(planes, group=None):
    """Optional full-block gather: planes[3][T][ncol_local] -> [world][3][T][ncol_local].
    Message size per rank is 3*T*ncol*elem bytes -- see DESIGN.md for why the statistics
    block, not this, is the default exchange."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return planes.unsqueeze(0)
    world = dist.get_world_size(group)
    return _gather0(planes, world, group)


# ---- particle-filter analysis step (BASELINE config C5) ---------------------------
def pf_systematic_ancestors(logw, u0, return_fixed=False, total_out=None):
    """Systematic resampling over the global particle set on the device (pf.hip):
    logw f64 CUDA tensor [n] -> int32 ancestors [n], non-decreasing, identical on every rank.
    With `total_out` (int64 CUDA tensor [1]) nothing is synchronised: the total integer weight
    is left there and the caller checks `total_out > 0` (a particle survived) later."""
    import ctypes as C
    import torch
    from ._lib import check, lib
    logw = logw.contiguous()
    assert logw.dtype == torch.float64 and logw.is_cuda
    anc = torch.empty(logw.numel(), dtype=torch.int32, device=logw.device)
    fixed = torch.empty(logw.numel(), dtype=torch.int64, device=logw.device) if return_fixed else None
    stream = C.c_void_p(torch.cuda.current_stream(logw.device).cuda_stream)
    if total_out is not None:
        assert total_out.dtype == torch.int64 and total_out.is_cuda
        check(lib().sipnet_pf_systematic_ancestors_async(
            C.c_void_p(logw.data_ptr()), logw.numel(), float(u0), C.c_void_p(anc.data_ptr()),
            C.c_void_p(fixed.data_ptr()) if return_fixed else None,
            C.c_void_p(total_out.data_ptr()), stream), "pf_systematic_ancestors")
    else:
        check(lib().sipnet_pf_systematic_ancestors(
            C.c_void_p(logw.data_ptr()), logw.numel(), float(u0), C.c_void_p(anc.data_ptr()),
            C.c_void_p(fixed.data_ptr()) if return_fixed else None, stream), "pf_systematic_ancestors")
    return (anc, fixed) if return_fixed else anc


def pf_exchange_plan(ancestors, n_local, world, rank):
    """Who sends what: from the global ancestor vector (identical on every rank) ->
    (send_cols[d]: local columns rank d needs from me, each once;
     src[n_local]: for my new column j, < n_local = my own old column, n_local + k = k-th
     received column; recv_counts[s]: columns arriving from rank s).
    Device tensors go through the library (sipnet_pf_exchange_plan: four kernels, one scan and
    ONE host synchronisation for the split sizes, whatever the number of ranks); the torch
    formulation below is the same plan for CPU tensors (the gloo tests) and its reference."""
    import torch
    if ancestors.is_cuda:
        import ctypes as C
        from ._lib import check, lib
        anc = ancestors.to(torch.int32).contiguous()
        send = torch.empty(world * n_local, dtype=torch.int32, device=anc.device)
        src = torch.empty(n_local, dtype=torch.int32, device=anc.device)
        sc, rc = (C.c_int64 * world)(), (C.c_int64 * world)()
        stream = C.c_void_p(torch.cuda.current_stream(anc.device).cuda_stream)
        check(lib().sipnet_pf_exchange_plan(C.c_void_p(anc.data_ptr()), n_local, world, rank,
                                            C.c_void_p(send.data_ptr()), C.c_void_p(src.data_ptr()),
                                            sc, rc, stream), "pf_exchange_plan")
        cols, off = [], 0
        for d in range(world):
            cols.append(send[off:off + sc[d]])
            off += sc[d]
        return cols, src, [int(x) for x in rc]
    return pf_exchange_plan_reference(ancestors, n_local, world, rank)


def pf_exchange_plan_reference(ancestors, n_local, world, rank):
    """the exchange plan in torch operations (any device)"""
    import torch
    n, lo = n_local, rank * n_local
    anc = ancestors.long()
    send_cols = []
    for d in range(world):
        if d == rank:
            send_cols.append(anc.new_empty(0, dtype=torch.int32))
            continue
        seg = anc[d * n:(d + 1) * n]
        sel = seg[(seg >= lo) & (seg < lo + n)]
        send_cols.append((torch.unique_consecutive(sel) - lo).to(torch.int32))
    mine = anc[lo:lo + n]
    owner = torch.div(mine, n, rounding_mode="floor")
    src = torch.empty(n, dtype=torch.int32, device=anc.device)
    recv_counts, start = [], 0
    for s in range(world):
        m = owner == s
        if s == rank:
            src[m] = (mine[m] - lo).to(torch.int32)
            recv_counts.append(0)
            continue
        u, inv = torch.unique_consecutive(mine[m], return_inverse=True)
        src[m] = (n + start + inv).to(torch.int32)
        recv_counts.append(int(u.numel()))
        start += int(u.numel())
    return send_cols, src, recv_counts


def pf_resample(batch, ancestors, rank=0, world=1, group=None, with_params=False, collectives=None):
    """Move particle state so that global particle g becomes old global particle
    ancestors[g]: local gather for ancestors this rank already holds, ONE all-to-all
    (RCCL over xGMI) of packed checkpoints for the rest, every needed ancestor sent once
    per destination.  `batch` needs ncol, pack_members(), resample().
    Returns {"sent": columns sent, "received": columns received, "bytes_sent": ...}.
    `collectives`: None = only with more than one rank; True = plan, pack and all-to-all even in a
    one-rank group (bench.py --force-dist runs the N-rank code path on one GPU that way)."""
    import torch
    import torch.distributed as dist
    n = batch.ncol
    if collectives is None:
        collectives = world > 1
    if not collectives:
        batch.resample(ancestors[:n].to(torch.int32), None, (), with_params)
        return {"sent": 0, "received": 0, "bytes_sent": 0}
    send_cols, src, recv_counts = pf_exchange_plan(ancestors, n, world, rank)
    blocks = [batch.pack_members(c, with_params) for c in send_cols]   # (an empty block is [words][0])
    words = blocks[0].shape[0]
    send = torch.cat([b.reshape(-1) for b in blocks])
    out_splits = [words * c for c in recv_counts]
    in_splits = [int(b.numel()) for b in blocks]
    if _host_staged(send, group):
        recv_h = torch.empty(sum(out_splits), dtype=send.dtype)
        dist.all_to_all_single(recv_h, send.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=group)
        recv = recv_h.to(send.device)
    else:
        recv = torch.empty(sum(out_splits), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    batch.resample(src, recv, recv_counts, with_params)
    sent = sum(int(c.numel()) for c in send_cols)
    return {"sent": sent, "received": sum(recv_counts), "bytes_sent": sent * words * 8}


def pf_analysis(batch, plane, obs, sigma, u0, rank=0, world=1, group=None, with_params=False,
                diagnostics=True, total_out=None, collectives=None):
    """One analysis step after a forecast: likelihood weights of this rank's particles ->
    all-gather of log-weights (n_total x 8 B) -> systematic resampling (redundant, identical
    on every rank) -> pf_resample.  Returns (ancestors, info).  `total_out` (int64 CUDA tensor
    [1]): run without a host round trip and leave the total weight there for a later check."""
    import torch
    import torch.distributed as dist
    if collectives is None:
        collectives = world > 1
    if not collectives and hasattr(batch, "pf_analysis_local"):
        # every particle is here: the three steps in one library call (without `diagnostics` the ancestors
        # returned are the batch's own buffer, overwritten by its next analysis)
        anc, logw = batch.pf_analysis_local(plane, obs, sigma, u0, with_params, total_out)
        if diagnostics:
            anc, logw = anc.clone(), logw.clone()
        info = {"sent": 0, "received": 0, "bytes_sent": 0}
    else:
        logw = batch.pf_log_weights(plane, obs, sigma)
        if collectives:
            logw = _gather0(logw, world, group).reshape(-1)
        anc = pf_systematic_ancestors(logw, u0, total_out=total_out)
        info = pf_resample(batch, anc, rank, world, group, with_params, collectives)
    if not diagnostics:
        return anc, info
    w = torch.exp(logw - logw.max())
    info["ess"] = float(w.sum() ** 2 / (w * w).sum())
    info["unique_ancestors"] = int(torch.unique_consecutive(anc).numel())
    return anc, info


# ---- the same analysis step without an all-to-all: peer reads (include/sipnet_amd.h, "the filter across ranks
# WITHOUT an all-to-all") ------------------------------------------------------------------------------------
def pf_connect_peers(batch, rank=0, world=1, group=None, with_params=True, pretend_world=0):
    """Once per filter: every rank publishes where its particles' checkpoint matrices live and maps the
    others' (hipIpc handles between processes).  Host-side, off the cycle.
    pretend_world = P > 1 (one real rank only): the slot count of a P-rank filter on one GPU -- the batch is connected to a
    world of P whose every member is itself (its own descriptor P times; it plays the last rank), so that the analysis goes
    over P x nmax weights and the gather reads "peers" that happen to be local: what an 8-GPU cycle costs apart from the
    links.  pf_arm_peers / pf_analysis_peers must be given the same pretend_world."""
    import torch.distributed as dist
    mine = batch.pf_publish(with_params)
    if pretend_world > 1:
        assert world == 1, "pretend_world: one real rank only"
        batch.pf_connect([mine] * pretend_world, pretend_world - 1)
        return
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, mine, group=group)
    else:
        every = [mine]
    batch.pf_connect(every, rank)


class DirectComm:
    """A RCCL communicator of the engine's own among ranks that are processes (sipnet_comm_*): collectives enqueued on the
    CALLER'S stream -- torch.distributed's process group runs its collectives on an internal stream, two cross-stream event
    waits away from the kernels around them (~10 us of a 140 us particle-filter cycle).  torch.distributed still launches the
    ranks and carries the 128-byte id from rank 0 to the others; the librccl is the one the process already holds (PyTorch's)."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from ._lib import lib, check
        import torch
        self.L = lib()
        # every rank asks for an id (only rank 0's is used): a rank whose RCCL is unusable finds out HERE, and all ranks agree on
        # that before any of them enters the collective ncclCommInitRank -- which would otherwise wait for the missing one
        ident = (C.c_uint8 * 128)()
        rc = self.L.sipnet_comm_unique_id(ident)
        if world > 1:
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=torch.device("cuda", int(device)))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                check(rc, "comm_unique_id")
                raise RuntimeError("DirectComm: another rank has no usable RCCL")
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * 128)(*box[0])
        else:
            check(rc, "comm_unique_id")
        h = C.c_void_p()
        check(self.L.sipnet_comm_create(ident, int(world), int(rank), int(device), C.byref(h)), "comm_create")
        self.h, self.world, self.rank = h, world, rank

    def all_gather(self, mine, gathered, stream):
        """gathered [world][L] (device, contiguous) <- every rank's `mine` [L]; `mine` may be gathered[rank] (in place)"""
        import ctypes as C
        from ._lib import check
        assert gathered.is_contiguous() and mine.is_contiguous() and gathered.shape[0] == self.world
        check(self.L.sipnet_comm_all_gather(self.h, C.c_void_p(mine.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                            mine.numel() * mine.element_size(), stream), "comm_all_gather")

    def close(self):
        if getattr(self, "h", None):
            self.L.sipnet_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pf_arm_peers(batch, obs, sigma, rank=0, world=1, pretend_world=0):
    """before the forecast's run() of a connected filter: that launch then leaves this rank's log-weights in its slice of
    the all-gather's buffer (sipnet_batch_pf_arm), and pf_analysis_peers only adds the block maxima"""
    import torch
    if pretend_world > 1:
        rank, world = pretend_world - 1, pretend_world
    L = batch.pf_block_len()
    gathered = getattr(batch, "_pf_gathered", None)
    if gathered is None or gathered.shape != (world, L):
        gathered = batch._pf_gathered = torch.empty((world, L), dtype=torch.float64, device=batch.device)
    batch.pf_arm_block(obs, sigma, gathered[rank])


def pf_analysis_peers(batch, plane, obs, sigma, u0, rank=0, world=1, group=None, total_out=None,
                      collectives=None, gathered=None, ancestors=None, diagnostics=False, pretend_world=0, comm=None):
    """One analysis step of a connected filter (pf_connect_peers): this rank's log-weight block -> ONE
    all-gather of the blocks -> weights, prefix sum and this rank's ancestors -> one gather that reads each
    ancestor where it lives.  No host synchronisation, no second collective.  Returns (ancestor slots
    int32 [ncol] -- rank * nmax + particle --, gathered blocks).  comm: a DirectComm -- the all-gather then goes
    through the engine's own RCCL communicator on the batch's stream instead of torch.distributed's process group."""
    import torch
    import torch.distributed as dist
    if collectives is None:
        collectives = world > 1
    pretend = pretend_world > 1
    if pretend:
        rank, world = pretend_world - 1, pretend_world
    L = batch.pf_block_len()
    if gathered is None:
        gathered = getattr(batch, "_pf_gathered", None)
        if gathered is None or gathered.shape != (world, L):
            gathered = batch._pf_gathered = torch.empty((world, L), dtype=torch.float64, device=batch.device)
    mine = gathered[rank]
    batch.pf_local_weights(plane, obs, sigma, mine)
    if pretend:
        # the one real rank's all-gather (in place, its own slice), then its block into the other ranks' slices: ONE device
        # copy of (P - 1) blocks stands where the links' time would be.  The blocks are identical, so every pretended rank
        # keeps exactly its own slots (no particle "crosses") and the ensemble evolves as the one-rank filter's does -- the
        # cycle is timed like for like against the plain one.  (Blocks shifted against each other make particles cross --
        # tools/pf_peers_time.py, tests/test_gpu_node.py do that -- but a filter fed rotated copies of itself selects another
        # ensemble: its forecasts ran up to 30 % longer from the phenology of the survivors alone, which says nothing about
        # the exchange.)
        if collectives and comm is not None:     # (the one real rank's communicator: world 1, in place)
            comm.all_gather(mine, mine.view(1, L), batch._stream())
        elif collectives:
            dist.all_gather_into_tensor(mine, mine, group=group)
        gathered[:rank].copy_(mine.expand(rank, L))
    elif collectives:
        if comm is not None:
            comm.all_gather(mine, gathered, batch._stream())
        elif _host_staged(mine, group):
            out = torch.empty((world, L), dtype=torch.float64)
            dist.all_gather_into_tensor(out.view(-1), mine.cpu(), group=group)
            gathered.copy_(out)
        else:   # in place: rank r's input is its own slice of the output
            dist.all_gather_into_tensor(gathered.view(-1), mine, group=group)
    anc = batch.pf_resample_peers(gathered, u0, ancestors, total_out)
    if not diagnostics:
        return anc, gathered
    nmax = (L * 256) // 257                      # L = nmax + ceil(nmax / 256)
    while nmax + (nmax + 255) // 256 < L:
        nmax += 1
    logw = gathered[:, :nmax].reshape(-1)
    w = torch.exp(logw - logw.max())
    crossed = int(((anc // nmax) != rank).sum())
    pinfo = batch.pf_info()
    # a crossing particle: state + ring + its 4-byte column in the replicated parameter bank -- or its parameter rows
    per = batch.L.sipnet_batch_member_words(batch.h, 0) * 8 + 4 if pinfo["params_by_index"] else batch.L.sipnet_batch_member_words(batch.h, 1) * 8
    info = {"ess": float(w.sum() ** 2 / (w * w).sum()), "unique_ancestors": int(torch.unique_consecutive(anc).numel()),
            "received": crossed, "sent": 0, "bytes_received": crossed * per, "bytes_per_crossing_particle": per,
            "exchange": "peer reads", "params_by_index": pinfo["params_by_index"], "analysis_one_launch": pinfo["fused"],
            "analysis_grid": pinfo["grid"], "analysis_slots": pinfo["n_slots"]}
    return anc, info
