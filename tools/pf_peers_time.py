#!/usr/bin/env python3
"""dev (GPU box): the cross-rank analysis (sipnet_batch_pf_resample_peers: weights + prefix sum + ancestors over ALL ranks'
slots in one launch, then the gather that reads every ancestor where it lives) at the slot count of a W-rank filter, on ONE
GPU: the batch is connected to a world of W whose every member is itself (sipnet_amd.dist.pf_connect_peers(pretend_world=W)),
the gathered buffer holds its block W times with shifted weights (so that particles cross "ranks").  ms per call (HIP events
over K calls) for W = 1, 2, 4, 8, with the parameters replicated (an index travels) and with SIPNET_KOPT_PF_MOVE_PARAMS (the
rows travel: round 5).  Run under `rocprofv3 --kernel-trace --stats` for the two kernels' own durations.
usage: pf_peers_time.py [n_particles_per_rank] [calls] [worlds, comma-separated]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import torch
import sipnet_amd as sa
from sipnet_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
T = 48
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
members = synth.perturbed_params(base, n)
for opt, tag in ((0, "index into the replicated bank"), (sa.KOPT_PF_MOVE_PARAMS, "parameter rows move")):
    for W in worlds:
        b = sa.Batch(flags, 1, n, sa.F32_MIXED, kernel=sa.KERNEL_ONE_WAVE, kernel_options=opt)
        b.set_climate(0, clim)
        b.set_params(0, members)
        b.setup()
        planes, _ = b.run(0, T)
        tot = planes[0].double().sum(0)
        obs, sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
        rank = W // 2
        d = b.pf_publish(with_params=True)
        b.pf_connect([d] * W, rank)
        L = b.pf_block_len()
        gathered = torch.empty((W, L), dtype=torch.float64, device=b.device)
        b.pf_local_weights(planes[0], obs, sigma, gathered[rank])
        mine = gathered[rank].clone()
        for r in range(W):
            gathered[r] = mine + 0.05 * (r - rank)
        total = torch.zeros(1, dtype=torch.int64, device=b.device)
        anc = torch.empty(n, dtype=torch.int32, device=b.device)
        for _ in range(10):
            b.pf_resample_peers(gathered, 0.5, anc, total)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(K):
            b.pf_resample_peers(gathered, 0.5, anc, total)
        e1.record()
        torch.cuda.synchronize()
        info = b.pf_info()
        print("world %d x %d particles (%d slots), %s: %.4f ms per resample_peers (one-launch analysis: %d, grid %d of budget %d; "
              "crossing %.0f per cycle), total weight %d" %
              (W, n, info["n_slots"], tag, e0.elapsed_time(e1) / K, info["fused"], info["grid"], info["budget"],
               info["crossing"] / max(info["cycles"], 1), int(total.item())), flush=True)
        if hasattr(sa.lib(), "sipnet_debug_read_pf_stamps"):      # a -DSIPNET_PF_STAMPS build: workgroup 0's clock (100 MHz) at every phase
            import ctypes as C
            st = (C.c_ulonglong * 8)()
            sa.lib().sipnet_debug_read_pf_stamps(st)
            t = [int(x) for x in st]
            print("   phases of the last launch, workgroup 0 (us): maxima %.2f | weights + block scan %.2f | barrier %.2f | chunk offsets %.2f | "
                  "ancestors %.2f" % tuple((t[j] - t[i]) / 100.0 for i, j in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6))), flush=True)
        b.close()
