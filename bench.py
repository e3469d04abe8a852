#!/usr/bin/env python3
"""bench.py -- ensemble-site-timesteps/s of the batched SIPNET step loop on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run, one rank per GPU.  One "step" = one pass
  of the hot path over the whole batch: per-member setup + the time-fused step kernel
  over every timestep of the forcing + (N>1) the ensemble-statistics reduction and the
  RCCL all-gather of the NEE/GPP/ET statistics block.  Inputs (parameters, site plan)
  are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  c10k  1 site x 10240 members, fp64, synthetic half-hourly year (17520 steps)  [default:
        the configuration the metric "at 10k members" + the fp64 |dNEE| bar are quoted on]
  c2    1 site x 1024 members, fp64          c3   1 site x 65536 members, fp32-mixed
  c4    32 sites x 1024 members per GPU, fp64 (256 sites over 8 GPUs)
  c5    particle-filter cycle: 131072 particles per GPU (1 M over 8), fp32-mixed, one day
        (48 steps) of forecast + the analysis step (likelihood weights, all-gather of
        log-weights, systematic resampling, all-to-all of resampled checkpoints, gather)
  c10kn c10k's shape with the nitrogen-cycle flag set (litter pool + anaerobic + N cycle): the
        optional-flag instantiation of the throughput kernel (not a BASELINE config)
Per-GPU work is fixed as N grows ("scaling": "weak").
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES = {"f64": 344.0, "f32": 172.0}   # bytes per member-timestep
HBM_PEAK_GBPS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    "c10k": dict(sites=1, members=10240, prec="f64", steps=17520),
    "c2": dict(sites=1, members=1024, prec="f64", steps=17520),
    "c3": dict(sites=1, members=65536, prec="f32", steps=17520),
    "c4": dict(sites=32, members=1024, prec="f64", steps=17520),
    "c5": dict(sites=1, members=131072, prec="f32", steps=48, pf=True),
    # c10k's shape with the nitrogen-cycle flag set (litter pool + anaerobic + N cycle)
    "c10kn": dict(sites=1, members=10240, prec="f64", steps=17520, param="allflags_forest.param",
                  flags=dict(litterPool=1, anaerobic=1, nitrogenCycle=1)),
}

_CPU_WORKER = r"""
import ctypes as C, numpy as np, sys, time, os
kind, so, param_file, clim_file, raw_path, flags_s = sys.argv[1:7]
flags = [int(x) for x in flags_s.split(',')]
raw = np.load(raw_path)
fl = (C.c_int*12)(*flags)
if kind == 'reference':
    ref = C.CDLL(so)
    n = ref.ref_init(fl, param_file.encode(), clim_file.encode(), b'/nonexistent', b'/dev/null')
    ref.ref_time_members.restype = C.c_double
    sink = C.c_double()
    dt = ref.ref_time_members(raw.ctypes.data_as(C.c_void_p), raw.shape[0], C.byref(sink))
else:
    sys.path.insert(0, os.environ['SIPNET_REPO'])
    import sipnet_amd as sa
    clim = sa.read_clim(clim_file)
    ora = C.CDLL(so)
    ora.sipo_time_members.restype = C.c_double
    sink = C.c_double()
    n = clim.n_steps
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    dt = ora.sipo_time_members(fl, vp(raw), raw.shape[0], n, vp(clim.data), vp(clim.year), vp(clim.day), C.byref(sink))
print(dt, raw.shape[0] * n)
"""


def usable_cores():
    """Host cores this process may really use: CPU affinity capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(flags, members_raw, raw_forcing, target_seconds=12.0, param_name="base_forest.param"):
    """Time the CPU checker (the real reference build when oracle/_ref travelled,
    else this repo's restatement) on a bounded sample of the same ensemble, one
    process per host core.  Test infrastructure used as a *baseline*, never shipped."""
    from sipnet_amd import synth
    ref_so = os.path.join(REPO, "oracle", "_ref", "libsipnet_ref.so")
    ora_so = os.path.join(REPO, "oracle", "liboracle.so")
    if os.path.exists(ref_so):
        kind, so = "reference", ref_so
    else:
        if not os.path.exists(ora_so):
            subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
        kind, so = "port", ora_so
    cores = usable_cores()
    n_steps = len(raw_forcing["year"])
    # ~350 ns per member-step per core (BASELINE.md probe) -> members per core
    per_core = max(1, int(target_seconds / (n_steps * 150e-9)))
    per_core = min(per_core, members_raw.shape[0] // cores if members_raw.shape[0] >= cores else 1)
    tmp = tempfile.mkdtemp(prefix="sipnet_cpu_")
    clim_file = os.path.join(tmp, "bench.clim")
    synth.write_clim(clim_file, raw_forcing)
    param_file = os.path.join(REPO, "sipnet_amd", "data", param_name)
    procs = []
    env = dict(os.environ, SIPNET_REPO=REPO)
    t0 = time.time()
    for c in range(cores):
        raw_path = os.path.join(tmp, f"raw{c}.npy")
        np.save(raw_path, np.ascontiguousarray(members_raw[c * per_core:(c + 1) * per_core]))
        procs.append(subprocess.Popen(
            [sys.executable, "-c", _CPU_WORKER, kind, so, param_file, clim_file, raw_path,
             ",".join(str(f) for f in flags)], stdout=subprocess.PIPE, env=env, text=True))
    secs, units = [], 0
    for p in procs:
        out = p.communicate()[0].strip().split()
        secs.append(float(out[0]))
        units += int(out[1])
    wall = time.time() - t0
    value = units / max(secs)
    return {
        "value": value, "unit": "ensemble-site-timesteps/s", "cores": cores, "kind": kind,
        "per_core": units / sum(secs),
        "sample": f"{per_core * cores} members x {n_steps} steps of the same synthetic ensemble, "
                  f"{cores} processes (one per host core), step loop only, gcc -O2; "
                  f"slowest process {max(secs):.2f}s, wall incl. start-up {wall:.1f}s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SIPNET_BENCH_WORKLOAD", "c10k"),
                    choices=sorted(WORKLOADS))
    ap.add_argument("--members", type=int, default=0, help="override members per site per GPU")
    ap.add_argument("--nsteps", type=int, default=0, help="override timesteps per pass")
    ap.add_argument("--fast-math", type=int, default=int(os.environ.get("SIPNET_FAST_MATH", "1")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="stats", choices=["stats", "full", "none"])
    ap.add_argument("--rehearse", action="store_true",
                    help="development: run the N>1 path on ONE GPU (all ranks share device 0, gloo "
                         "collectives through host copies); the numbers mean nothing")
    args = ap.parse_args()

    wl = dict(WORKLOADS[args.workload])
    if args.members:
        wl["members"] = args.members
    if args.nsteps:
        wl["steps"] = args.nsteps
    os.environ["SIPNET_FAST_MATH"] = str(args.fast_math)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE={world})",
              file=sys.stderr)
        sys.exit(2)

    import sipnet_amd as sa
    from sipnet_amd import synth

    flags = sa.flags_from(**wl.get("flags", {}))
    param_name = wl.get("param", "base_forest.param")
    base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", param_name), flags)
    S, M, T = wl["sites"], wl["members"], wl["steps"]
    prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED

    # identical inputs for CPU baseline and GPU: rank r owns global members [r*M, (r+1)*M)
    # of each of its sites; sites of rank r are r*S .. r*S+S-1
    raws = [synth.round_like_file(synth.half_hourly_year_raw(T, site=rank * S + s)) for s in range(S)]
    clims = [synth.convert_raw(r) for r in raws]
    members = synth.perturbed_params(base, M * world, seed=synth.SEED_PARAMS)[rank * M:(rank + 1) * M]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # before any HIP initialisation in this process: the workers are plain children
        cpu = cpu_baseline(flags, synth.perturbed_params(base, max(M, 64), seed=synth.SEED_PARAMS),
                           raws[0], param_name=param_name)

    import torch
    import torch.distributed as dist
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def all_gather_into(out, x):
        if args.rehearse:   # gloo: through the host
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o.view((-1,) + tuple(x.shape[1:])) if x.dim() else o, x.cpu().contiguous())
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, x)

    b = sa.Batch(flags, S, M, prec, device=local_rank, fast_math=bool(args.fast_math))
    for s in range(S):
        b.set_climate(s, clims[s])
        b.set_params(s, members)
    b.setup()
    planes, _ = b.alloc_outputs(T)
    stats = torch.empty((3, T, S, 2), dtype=torch.float64, device=b.device)
    gathered = torch.empty((world,) + tuple(stats.shape), dtype=torch.float64, device=b.device) \
        if world > 1 and args.gather == "stats" else None
    gathered_full = None
    if world > 1 and args.gather == "full":
        gathered_full = torch.empty((world,) + tuple(planes.shape), dtype=planes.dtype, device=b.device)

    # N > 1, statistics gather: the ensemble statistics of pass k (three streaming reductions
    # + one small all-gather) run on a side stream under the step kernel of pass k+1, which
    # writes the other of two output-plane buffers
    overlap = world > 1 and args.gather == "stats" and not wl.get("pf")
    if overlap:
        planes2, _ = b.alloc_outputs(T)
        stats2 = torch.empty_like(stats)
        gathered2 = torch.empty_like(gathered)
        side = torch.cuda.Stream(device=b.device)
        bufs = [dict(planes=planes, stats=stats, gathered=gathered, ran=torch.cuda.Event(), done=None),
                dict(planes=planes2, stats=stats2, gathered=gathered2, ran=torch.cuda.Event(), done=None)]
        npass = [0]

    kernel_ms = []
    pf = bool(wl.get("pf"))
    pf_info = {}
    if pf:
        # the "observation": the ensemble's median daily NEE with a spread that leaves an
        # effective sample size of roughly half the ensemble
        from sipnet_amd import dist as sd
        b.run(0, T, planes=planes)
        tot = planes[0].double().sum(0)
        if world > 1:
            tot = sd._gather0(tot, world, None).reshape(-1)
        pf_obs, pf_sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12

    pf_totals = torch.ones(max(args.steps + args.warmup, 1), dtype=torch.int64, device=b.device)
    pf_cycle = [0]

    def one_pass(record):
        if overlap:
            buf = bufs[npass[0] & 1]
            npass[0] += 1
            main = torch.cuda.current_stream()
            if buf["done"] is not None:
                main.wait_event(buf["done"])       # its statistics (two passes ago) have left
            b.setup()
            b.run(0, T, planes=buf["planes"])
            buf["ran"].record(main)
            with torch.cuda.stream(side):
                side.wait_event(buf["ran"])
                for v in range(3):
                    b.reduce_plane(buf["planes"][v], buf["stats"][v])
                all_gather_into(buf["gathered"], buf["stats"])
                buf["done"] = torch.cuda.Event()
                buf["done"].record(side)
            return
        b.setup()                       # setupModel() for every member
        b.run(0, T, planes=planes)      # the time-fused step kernel
        if pf:
            # no host round trip inside a cycle: each cycle's total weight lands in its own slot
            # and "a particle survived" is checked for all cycles after the closing barrier
            slot = pf_totals[pf_cycle[0] % len(pf_totals):][:1]
            pf_cycle[0] += 1
            _, info = sd.pf_analysis(b, planes[0], pf_obs, pf_sigma, u0=0.5, rank=rank, world=world,
                                     with_params=True, diagnostics=record, total_out=slot)
            pf_info.update(info)
            return
        if world > 1 and args.gather != "none":
            for v in range(3):
                b.reduce_plane(planes[v], stats[v])
            if args.gather == "stats":
                all_gather_into(gathered, stats)
            else:
                all_gather_into(gathered_full, planes)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass(True)
    barrier()
    # HIP-event kernel timing is collected in an extra, untimed pass per step to keep the
    # timed region free of host syncs: time K passes wall-clock first
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(False)
    barrier()
    dt = time.perf_counter() - t0
    if pf and not bool((pf_totals > 0).all()):
        raise RuntimeError("particle filter: a cycle ended with every particle at zero weight")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else b.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # dominant kernel's launch duration, HIP events on the launch stream
    kms = []
    for _ in range(max(3, min(args.steps, 5))):
        b.setup()
        b.run(0, T, planes=planes)
        kms.append(b.last_kernel_ms())
    k_ms = float(np.mean(kms))
    if pf:   # the analysis step alone, torch events on the current stream (all its work is there)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ams = []
        for _ in range(3):
            b.setup()
            b.run(0, T, planes=planes)
            barrier()
            e0.record()
            sd.pf_analysis(b, planes[0], pf_obs, pf_sigma, u0=0.5, rank=rank, world=world, with_params=True,
                           diagnostics=False)
            e1.record()
            torch.cuda.synchronize()
            ams.append(e0.elapsed_time(e1))
        pf_info["analysis_ms"] = float(np.mean(ams))

    units_per_pass = S * M * T * world
    value = units_per_pass * args.steps / dt
    per_launch_units = S * M * T
    achieved = ALGO_BYTES[wl["prec"]] * per_launch_units / (k_ms * 1e-3) / 1e9

    # parity spot-check of this very run against the CPU oracle (checker only)
    parity = None
    if rank == 0:
        try:
            from tests import helpers
            ora = helpers.load_oracle()
            n_chk = min(8, M)
            po, _, _ = ora.run_block(flags, members[:n_chk], clims[0])
            if pf:      # the planes of the last forecast (the analysis does not touch them)
                b.setup()
                b.run(0, T, planes=planes)
            pg = planes[:, :, :n_chk].double().cpu().numpy()
            parity = {"members_checked": n_chk,
                      "max_abs_dNEE": float(np.abs(pg[0] - po[0]).max()),
                      "max_abs_dGPP": float(np.abs(pg[1] - po[1]).max()),
                      "max_abs_dET": float(np.abs(pg[2] - po[2]).max())}
        except Exception as e:  # the checker is optional for the measurement itself
            parity = {"error": repr(e)}

    traffic = None
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "ensemble-site-timesteps/sec", "value": value,
            "unit": "ensemble-site-timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl["prec"], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {S} site(s) x {M} members per GPU x {T} "
                                   f"half-hourly steps, perturbed params, " + ("default flags" if not wl.get("flags") else "flags " + "+".join(sorted(wl["flags"])))
                                   + (", particle-filter cycle (forecast + analysis)" if pf else ""),
                       "sites_per_gpu": S, "members_per_site": M, "timesteps": T,
                       "fast_math": bool(args.fast_math),
                       "gather": args.gather if world > 1 else "n/a (1 GPU)",
                       "parallelism": f"ensemble-sharded x{world}",
                       **({"particle_filter": pf_info} if pf else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("stepCoopKernel" if S * ((M + 63) // 64) <= 512 and os.environ.get("SIPNET_COOP", "1") != "0" and not wl.get("flags") else "stepFastKernel") if args.fast_math else "stepKernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_unit": ALGO_BYTES[wl["prec"]],
                         "units_per_launch": per_launch_units},
            "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    b.close()


if __name__ == "__main__":
    main()
