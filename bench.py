#!/usr/bin/env python3
"""bench.py -- ensemble-site-timesteps/s of the batched SIPNET step loop on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1: one rank per GPU under torch.distributed.run -- started by the driver, or, when the command is given
  without a launcher (no WORLD_SIZE in the environment), by bench.py itself as a child process (launch_ranks).  One "step" = one pass
  of the hot path over the whole batch: per-member setup + the time-fused step kernel
  over every timestep of the forcing + (N>1) the ensemble statistics (summed inside the
  step kernel's launch) and the RCCL all-gather of the NEE/GPP/ET statistics block.  Inputs (parameters, site plan)
  are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  c10k  1 site x 10240 members, fp64, synthetic half-hourly year (17520 steps)  [default:
        the configuration the metric "at 10k members" + the fp64 |dNEE| bar are quoted on]
  c2    1 site x 1024 members, fp64          c3   1 site x 65536 members, fp32-mixed
  c4    32 sites x 1024 members per GPU, fp64 (256 sites over 8 GPUs; rank r owns sites 32r..32r+31)
  c5    particle-filter cycle: 131072 particles per GPU (1 M over 8), fp32-mixed, one day
        (48 steps) of forecast + the analysis step (likelihood weights, ONE all-gather of the
        log-weight blocks, systematic resampling of the rank's own particles over the gathered
        weights, one gather that reads every ancestor where it lives -- peer HBM over xGMI;
        --pf-exchange alltoall: the all-to-all of packed checkpoints instead)
  c2x16 16 sites x 1024 members per GPU, fp64: c2 stacked 16-fold (INTEGRATION.md "small ensembles")
  c4n   c4's shape with the nitrogen-cycle flag set (two chunks per CU)
  c10kn c10k's shape with the nitrogen-cycle flag set (litter pool + anaerobic + N cycle): the
        optional-flag instantiation of the throughput kernel (not a BASELINE config)
  c10kr3 c10k's shape with russell_3's flags (growth respiration + leaf water + litter pool): the
        optional-physics instantiation of the cooperative kernel (not a BASELINE config)
Per-GPU work is fixed as N grows ("scaling": "weak").

What the JSON line says about the kernel (the `roofline` object):
  frac               SURVEY 8(d)'s contract number: 344 B (172 B fp32) of ALGORITHMIC state traffic
                     per member-step x units per launch / kernel time / 8 TB/s.  It is a normalised
                     throughput: the time-fused kernel keeps state in registers and LDS, so ...
  hbm_measured_frac  ... the bytes that really cross HBM (rocprofv3 PMC, profiles/pmc_traffic.json,
                     tagged with the round they were collected in) / kernel time / 8 TB/s;
  waves_per_simd, cus_used, simds_used   the launch shape (from the library, not re-derived here);
  valu_busy_frac     the physical ceiling of a time-fused fp64 kernel (SURVEY 8(d) "ALU cross-check"): SIMD-cycles
                     with the vector ALU executing (SQ_ACTIVE_INST_VALU x 4, committed profile) / (all SIMDs x
                     kernel time x 2.4 GHz); valu_busy_frac_of_used_simds: the same over the SIMDs the launch occupies;
  issue_frac         this run's rate / the rate the same model reaches on this GPU once every SIMD
                     holds two wavefronts (a short 131 072-member probe of the one-wave kernel, run
                     untimed in this process): how much of the chip's instruction issue the launch uses;
  end_to_end         the whole job with the PCIe legs (parameters + climate up, plan, setup, kernel with
                     statistics, statistics down), measured untimed in the same run: never `value`;
  plan_ms, setup_ms  host-side site-plan build + upload (once per forcing, before the timed region;
                     measured on a second hand-over of the same climate, the first one in a
                     process also pays the runtime's first-use costs: plan_ms_first_in_process)
                     and the per-pass setupModel() kernel (inside it).
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
SCLK_PEAK_HZ = 2.4e9                         # MI355X_MICROARCH.md: max clock 2400 MHz

WORKLOADS = {
    "c10k": dict(sites=1, members=10240, prec="f64", steps=17520),
    "c2": dict(sites=1, members=1024, prec="f64", steps=17520),
    "c3": dict(sites=1, members=65536, prec="f32", steps=17520),
    "c4": dict(sites=32, members=1024, prec="f64", steps=17520),
    # what a caller with a SMALL ensemble should hand over: 16 sites (or years / scenarios) of
    # 1 024 members stacked into one batch -- 256 chunks, one per CU, for the launch time of c2
    "c2x16": dict(sites=16, members=1024, prec="f64", steps=17520),
    "c5": dict(sites=1, members=131072, prec="f32", steps=48, pf=True),
    # c10k's shape with the nitrogen-cycle flag set (litter pool + anaerobic + N cycle)
    "c10kn": dict(sites=1, members=10240, prec="f64", steps=17520, param="allflags_forest.param",
                  flags=dict(litterPool=1, anaerobic=1, nitrogenCycle=1)),
    # c10k's shape with russell_3's flag family (growth respiration + leaf water + litter pool, no moisture effect on
    # heterotrophic respiration): the optional-physics instantiation of the one-chunk cooperative kernel (run-time flags)
    "c10kr3": dict(sites=1, members=10240, prec="f64", steps=17520, param="allflags_forest.param",
                   flags=dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0)),
    # ... and c4's (two chunks per CU: the two-chunk layout of that kernel)
    "c4n": dict(sites=32, members=1024, prec="f64", steps=17520, param="allflags_forest.param",
                flags=dict(litterPool=1, anaerobic=1, nitrogenCycle=1)),
}

_CPU_WORKER = r"""
import ctypes as C, numpy as np, sys, time, os
kind, so, param_file, clim_file, raw_path, flags_s, cpu = sys.argv[1:8]
if int(cpu) >= 0:
    os.sched_setaffinity(0, {int(cpu)})          # one worker pinned to one host core
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


def usable_cpus():
    """Host cores this process may really use: the CPU affinity list capped by the cgroup CPU
    quota (the GPU box shows 256 logical CPUs but grants a 16-CPU quota)."""
    cpus = sorted(os.sched_getaffinity(0))
    n = len(cpus)
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
    # spread the pinned workers over the allowed set (neighbouring logical CPUs tend to be the
    # busiest ones of a shared host, and may be SMT siblings)
    stride = max(1, len(cpus) // n)
    return cpus[::stride][:n]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_leg(kind, so, flags, members_raw, n_steps, clim_file, param_file, cpus, target_seconds, ns_guess):
    """one timed leg: len(cpus) worker processes, each pinned to its own core (cpu = -1: left
    to the host's scheduler)"""
    cores = len(cpus)
    per_core = max(1, int(target_seconds / (n_steps * ns_guess * 1e-9)))
    per_core = min(per_core, members_raw.shape[0] // cores if members_raw.shape[0] >= cores else 1)
    tmp = os.path.dirname(clim_file)
    procs = []
    env = dict(os.environ, SIPNET_REPO=REPO)
    t0 = time.time()
    for c, cpu in enumerate(cpus):
        raw_path = os.path.join(tmp, f"raw{c}.npy")
        np.save(raw_path, np.ascontiguousarray(members_raw[c * per_core:(c + 1) * per_core]))
        procs.append(subprocess.Popen(
            [sys.executable, "-c", _CPU_WORKER, kind, so, param_file, clim_file, raw_path,
             ",".join(str(f) for f in flags), str(cpu)], stdout=subprocess.PIPE, env=env, text=True))
    secs, units = [], 0
    for p in procs:
        out = p.communicate()[0].strip().split()
        secs.append(float(out[0]))
        units += int(out[1])
    return dict(value=units / max(secs), per_core=units / sum(secs), members=per_core * cores,
                slowest_s=max(secs), wall_s=time.time() - t0)


def cpu_baseline(flags, members_raw, raw_forcing, param_name="base_forest.param"):
    """Time the CPU checker (the real reference build when oracle/_ref travelled, else this
    repo's restatement) on a bounded sample of the same ensemble, one PINNED process per usable
    host core, at gcc -O2 (`value`) and at -O0, the level the reference's own Makefile ships
    (`value_O0`).  Test infrastructure used as a *baseline*, never shipped."""
    from sipnet_amd import synth
    ref_so = os.path.join(REPO, "oracle", "_ref", "libsipnet_ref.so")
    ref_o0 = os.path.join(REPO, "oracle", "_ref", "libsipnet_ref_O0.so")
    ora_so = os.path.join(REPO, "oracle", "liboracle.so")
    if os.path.exists(ref_so):
        kind, so = "reference", ref_so
    else:
        if not os.path.exists(ora_so):
            subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "oracle"])
        kind, so = "port", ora_so
    cpus = usable_cpus()
    n_steps = len(raw_forcing["year"])
    tmp = tempfile.mkdtemp(prefix="sipnet_cpu_")
    clim_file = os.path.join(tmp, "bench.clim")
    synth.write_clim(clim_file, raw_forcing)
    param_file = os.path.join(REPO, "sipnet_amd", "data", param_name)
    # the box is shared: which cores are quiet differs from run to run, so the -O2 leg runs twice --
    # pinned (one worker per core of a spread-out set) and unpinned (the host scheduler places the
    # workers) -- and the FASTER of the two is the baseline's value; both are reported
    pin = _cpu_leg(kind, so, flags, members_raw, n_steps, clim_file, param_file, cpus, 6.0, 150)
    free = _cpu_leg(kind, so, flags, members_raw, n_steps, clim_file, param_file, [-1] * len(cpus), 6.0, 150)
    o2 = pin if pin["value"] >= free["value"] else free
    o0 = None
    if kind == "reference" and os.path.exists(ref_o0):
        o0 = _cpu_leg(kind, ref_o0, flags, members_raw, n_steps, clim_file, param_file,
                      cpus if o2 is pin else [-1] * len(cpus), 5.0, 400)
    return {
        "value": o2["value"], "unit": "ensemble-site-timesteps/s", "cores": len(cpus), "kind": kind,
        "per_core": o2["per_core"], "pinned": o2 is pin, "value_pinned": pin["value"],
        "value_unpinned": free["value"], "cpu_model": cpu_model(),
        "value_O0": o0["value"] if o0 else None, "per_core_O0": o0["per_core"] if o0 else None,
        "sample": f"{o2['members']} members x {n_steps} steps of the same synthetic ensemble, "
                  f"{len(cpus)} processes (one per usable host core; pinned {pin['value'] / 1e6:.1f} M/s, "
                  f"unpinned {free['value'] / 1e6:.1f} M/s, the faster one is `value`), step loop only, gcc -O2 "
                  f"(slowest process {o2['slowest_s']:.2f}s)"
                  + (f"; the same at -O0 (the reference Makefile's level) on {o0['members']} members "
                     f"(slowest {o0['slowest_s']:.2f}s)" if o0 else ""),
    }


def fill_probe(sa, synth, flags, base, prec_name, device):
    """Rate of the same model on this GPU with every SIMD holding two wavefronts: 131 072 members
    of the one-wave throughput kernel over 960 steps (NEE plane only).  Untimed extra; gives
    `issue_frac` its denominator."""
    import torch
    M, T = 131072, 960
    prec = sa.F64 if prec_name == "f64" else sa.F32_MIXED
    b = sa.Batch(flags, 1, M, prec, device=device, fast_math=True if prec == sa.F64 else None,
                 kernel=sa.KERNEL_ONE_WAVE)
    b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
    b.set_params(0, synth.perturbed_params(base, M, seed=synth.SEED_PARAMS))
    nee = torch.empty((1, T, M), dtype=b.out_dtype, device=b.device)
    import ctypes as C
    ms = []
    for _ in range(3):
        b.setup()
        sa._lib.check(b.L.sipnet_batch_run(b.h, 0, T, C.c_void_p(nee.data_ptr()), None, None, None,
                                           M, b._stream()), "run")
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    li = b.last_launch()
    b.close()
    return {"rate": M * T / (min(ms) * 1e-3), "kernel": li["kernel"], "members": M, "timesteps": T,
            "kernel_ms": min(ms), "waves_per_simd": li["waves_per_simd"]}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher in front: start the N ranks ourselves, exactly as the driver's
    N > 1 command does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>`), as a CHILD process -- this process has neither imported torch nor
    touched HIP, and it never replaces itself (an exec from a process that has initialised the GPU takes the machine
    down on this pool).  The child inherits stdout, so rank 0's one JSON line is this command's one JSON line; the
    return code is the child's.  The reference's model is one run = one process (frontend.c:130-253): the launcher
    multiplies that."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:      # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL and the peer-mapped checkpoints need it
    env.setdefault("OMP_NUM_THREADS", "1")               # (what torch.distributed.run would set, without its warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    try:
        return subprocess.call(cmd, env=env)
    except KeyboardInterrupt:
        return 130


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("SIPNET_BENCH_WORKLOAD", "c10k"),
                    choices=sorted(WORKLOADS))
    ap.add_argument("--members", type=int, default=0, help="override members per site per GPU")
    ap.add_argument("--nsteps", type=int, default=0, help="override timesteps per pass")
    ap.add_argument("--fast-math", type=int, default=1,
                    help="1: throughput kernels (default, what the metric is quoted on); 0: strict-order kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fill-probe", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--gather", default="stats", choices=["stats", "sums", "full", "none"],
                    help="N > 1, what every rank all-gathers per pass: stats = the ensemble statistics block (default), sums = "
                         "every member's NEE/GPP/ET sums over --sum-steps steps, summed inside the step kernel's launch "
                         "(sipnet_batch_run_sums) and gathered under the next pass, full = the member-resolved planes")
    ap.add_argument("--sum-steps", type=int, default=48, help="--gather sums: steps per group (48 = daily sums of a half-hourly year)")
    ap.add_argument("--dump-stats", default="",
                    help="rank 0 writes the whole ensemble's statistics block [3][T][sites][2] (sum, sum of "
                         "squares over ALL ranks' members) of the last pass to this .npy file")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 path (RCCL process group, side-stream statistics + all-gather, segmented "
                         "full gather, the particle filter's all-gather / all-to-all) even with ONE rank, and "
                         "report its cost against the plain pass as config.dist_overhead_ms")
    ap.add_argument("--pf-collective", default="direct", choices=["direct", "torch"],
                    help="c5, N > 1, peer exchange: the all-gather of log-weight blocks through the engine's own RCCL communicator on the "
                         "batch's stream (sipnet_comm_*; default, falls back to torch when it cannot be created on every rank) or "
                         "through torch.distributed's process group (its internal stream: two cross-stream waits per cycle)")
    ap.add_argument("--pf-exchange", default="peer", choices=["peer", "alltoall"],
                    help="c5, N > 1: how resampled particles cross ranks -- peer: ONE all-gather of log-weight blocks, "
                         "then every rank reads its ancestors straight out of its peers' HBM (IPC-mapped, xGMI); "
                         "alltoall: all-gather + exchange plan (one host round trip) + all_to_all_single of packed checkpoints")
    ap.add_argument("--rehearse", action="store_true",
                    help="development: run the N>1 path on ONE GPU (all ranks share device 0, gloo "
                         "collectives through host copies); the numbers mean nothing")
    ap.add_argument("--pretend-world", type=int, default=0,
                    help="c5 with --force-dist on ONE rank: run the exchange path at the slot count of a filter of this "
                         "many ranks (the batch connected to a world whose every member is itself: weights, prefix sum and "
                         "ancestors over P x 131 072 slots, the gather reading 'peers' that are local) -- what an 8-GPU "
                         "cycle costs apart from the links' own time")
    ap.add_argument("--launch-probe", action="store_true",
                    help="development / CPU test: only check that the N ranks come up and can talk (gloo, no GPU "
                         "touched): rank 0 prints {\"launch_probe\": true, \"ranks_seen\": N}")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: be our own launcher
        sys.exit(launch_ranks(args.gpus))
    if args.launch_probe:
        import torch
        import torch.distributed as dist
        if "WORLD_SIZE" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("gloo")
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, (dist.get_rank(), os.getpid()))
        if dist.get_rank() == 0:
            print(json.dumps({"launch_probe": True, "n_gpus": args.gpus, "ranks_seen": len({r for r, _ in got}),
                              "processes_seen": len({p for _, p in got})}), flush=True)
        dist.destroy_process_group()
        return

    wl = dict(WORKLOADS[args.workload])
    if args.members:
        wl["members"] = args.members
    if args.nsteps:
        wl["steps"] = args.nsteps

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} under a launcher that started {world} ranks (WORLD_SIZE={world})",
              file=sys.stderr)
        sys.exit(2)

    import sipnet_amd as sa
    from sipnet_amd import synth

    flags = sa.flags_from(**wl.get("flags", {}))
    param_name = wl.get("param", "base_forest.param")
    base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", param_name), flags)
    S, M, T = wl["sites"], wl["members"], wl["steps"]
    prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED

    # identical inputs for CPU baseline and GPU.  One-site workloads shard the MEMBER axis: every
    # rank runs site 0 with global members [r*M, (r+1)*M).  c4 shards whole SITES: rank r owns
    # sites r*S .. r*S+S-1 (and uploads only their plans), each with members [r*M, (r+1)*M) of the draw.
    site_ids = [rank * S + s for s in range(S)] if S > 1 else [0]
    # a particle filter advances: cycle k forecasts day k of the forcing from the resampled state (no setupModel()
    # inside a cycle); the forcing holds as many days as the run has cycles (wrapping, with a fresh setup, beyond that)
    pf_days = min(365, max(64, args.warmup + args.steps + 48)) if wl.get("pf") else 1
    raws = [synth.round_like_file(synth.half_hourly_year_raw(T * pf_days, site=sid)) for sid in site_ids]
    clims = [synth.convert_raw(r) for r in raws]
    members = synth.perturbed_params(base, M * world, seed=synth.SEED_PARAMS)[rank * M:(rank + 1) * M]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # before any HIP initialisation in this process: the workers are plain children
        cpu = cpu_baseline(flags, synth.perturbed_params(base, max(M, 64), seed=synth.SEED_PARAMS),
                           {k: v[:T] for k, v in raws[0].items()}, param_name=param_name)

    import torch
    import torch.distributed as dist
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # distd: the exchange path of an N-rank run is executed (N > 1, or --force-dist with one rank:
    # RCCL itself, the side stream and every collective call run exactly as they do at N = 8)
    distd = world > 1 or args.force_dist
    if distd:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            if world == 1 and "MASTER_ADDR" not in os.environ:     # started without the launcher
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"),
                                  RANK="0", WORLD_SIZE="1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def all_gather_into(out, x):
        """out[world, *x.shape] <- every rank's x; passed to the collective in its concatenated
        form (world * x.shape[0], ...), which RCCL and gloo both accept"""
        flat = (world * x.shape[0],) + tuple(x.shape[1:])
        if args.rehearse:   # gloo: through the host
            o = torch.empty(flat, dtype=out.dtype)
            dist.all_gather_into_tensor(o, x.cpu().contiguous())
            out.copy_(o.view(out.shape))
        else:
            dist.all_gather_into_tensor(out.view(flat), x.contiguous())

    # who takes part: every rank's device identity, gathered (proof that N ranks drove N devices)
    ranks_seen, devices_seen = 1, None
    props = torch.cuda.get_device_properties(local_rank)
    ident = f"{getattr(props, 'uuid', '')}|{getattr(props, 'pci_bus_id', '')}|{getattr(props, 'pci_device_id', '')}|{props.name}"
    if distd:
        me = torch.zeros(128, dtype=torch.uint8)
        raw_id = ident.encode()[:128]
        me[:len(raw_id)] = torch.tensor(list(raw_id), dtype=torch.uint8)
        allid = torch.zeros(world * 128, dtype=torch.uint8, device="cpu" if args.rehearse else torch.device("cuda", local_rank))
        dist.all_gather_into_tensor(allid, me.to(allid.device))
        ids = [bytes(allid[r * 128:(r + 1) * 128].cpu().tolist()).rstrip(b"\0").decode(errors="replace")
               for r in range(world)]
        ranks_seen, devices_seen = len(ids), len(set(ids))
        device_ids = ids
    else:
        device_ids = [ident]
        devices_seen = 1

    b = sa.Batch(flags, S, M, prec, device=local_rank, fast_math=bool(args.fast_math) if prec == sa.F64 else None)
    for s in range(S):
        b.set_climate(s, clims[s])
        b.set_params(s, members)
    b.setup()                       # builds and uploads the site plans (host side; timed by the library)
    li0 = b.last_launch()
    plan_ms = None                  # read after the first launch: the per-step records upload on first use
    planes, _ = b.alloc_outputs(T)
    stats = torch.empty((3, T, S, 2), dtype=torch.float64, device=b.device)
    gathered = torch.empty((world,) + tuple(stats.shape), dtype=torch.float64, device=b.device) \
        if distd and args.gather == "stats" else None
    gathered_full = None
    if distd and args.gather == "full":
        gathered_full = torch.empty((world,) + tuple(planes.shape), dtype=planes.dtype, device=b.device)

    # N > 1, statistics gather: the ensemble statistics of pass k come out of the step kernel's own
    # launch (sipnet_batch_run_stats); their all-gather runs on a side stream under the step kernel
    # of pass k+1, which writes the other of two output-plane / statistics buffers
    overlap = distd and args.gather in ("stats", "sums") and not wl.get("pf")
    side = torch.cuda.Stream(device=b.device) if distd else None
    sum_groups = (T + args.sum_steps - 1) // args.sum_steps
    if overlap and args.gather == "sums":
        if not b.sums_in_kernel():
            raise SystemExit("--gather sums: this batch's kernel has no in-launch sums (every throughput kernel has; the strict-order one "
                             "has not); the C host's sipnet_node_run_gathering_reduced sums such planes on its second stream")
        bufs = [dict(sums=torch.empty((3, sum_groups, b.ncol), dtype=torch.float64, device=b.device),
                     gathered=torch.empty((world, 3, sum_groups, b.ncol), dtype=torch.float64, device=b.device),
                     ran=torch.cuda.Event(), done=None) for _ in range(2)]
        npass = [0]
    elif overlap:
        planes2, _ = b.alloc_outputs(T)
        stats2 = torch.empty_like(stats)
        gathered2 = torch.empty_like(gathered)
        bufs = [dict(planes=planes, stats=stats, gathered=gathered, ran=torch.cuda.Event(), done=None),
                dict(planes=planes2, stats=stats2, gathered=gathered2, ran=torch.cuda.Event(), done=None)]
        npass = [0]

    pf = bool(wl.get("pf"))
    pf_info = {}
    if pf:
        # the "observation": the ensemble's median daily NEE with a spread that leaves an
        # effective sample size of roughly half the ensemble
        from sipnet_amd import dist as sd
        b.run(0, T, planes=planes)
        # the parity sample is taken from this first forecast: later cycles run RESAMPLED
        # parameter sets (particles carry their parameters), member k is no longer draw k
        pf_first = planes[:, :, :min(8, M)].double().cpu().numpy()
        tot = planes[0].double().sum(0)
        if distd:
            tot = sd._gather0(tot, world, None).reshape(-1)
        pf_obs, pf_sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
        # the observation series: what the free-running ensemble says on each day (untimed pre-pass), so that the
        # weights stay as informative on day k as on day 0
        pf_obs_k, pf_sigma_k = [pf_obs], [pf_sigma]
        for k in range(1, pf_days):
            b.run(k * T, T, planes=planes)
            tot = planes[0].double().sum(0)
            if distd:
                tot = sd._gather0(tot, world, None).reshape(-1)
            pf_obs_k.append(float(tot.median()))
            pf_sigma_k.append(float(tot.std()) * 1.5 + 1e-12)
        b.setup()
        pretend = args.pretend_world if (distd and world == 1 and args.pf_exchange == "peer") else 0
        pf_exchange = args.pf_exchange if distd else "n/a (1 GPU)"
        if distd and args.pf_exchange == "peer":
            try:    # once per filter, off the cycle: publish / map the checkpoint matrices of every rank
                sd.pf_connect_peers(b, rank, world, with_params=True, pretend_world=pretend)
            except Exception as e:   # (no IPC between these processes: the all-to-all path is the other product path)
                pf_exchange = f"alltoall (peer mapping failed: {e!r})"
            if world > 1:            # every rank takes the same path
                ok = torch.tensor([1 if pf_exchange == "peer" else 0], dtype=torch.int32,
                                  device="cpu" if args.rehearse else b.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and pf_exchange == "peer":
                    pf_exchange = "alltoall (a peer could not map)"

    # the filter's one collective on the batch's own stream: a RCCL communicator of the engine's (sipnet_comm_*), created once;
    # every rank takes the same path (any rank that cannot create it sends all of them back to torch.distributed's group)
    pf_comm, pf_collective = None, "n/a"
    if pf and distd and pf_exchange == "peer":
        pf_collective = "torch.distributed process group"
        if args.pf_collective == "direct" and not args.rehearse:
            why = None
            try:
                pf_comm = sd.DirectComm(rank, world, local_rank)
            except Exception as e:     # noqa: BLE001
                why = repr(e)[:200]
            if world > 1:
                ok = torch.tensor([1 if pf_comm is not None else 0], dtype=torch.int32, device=b.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and pf_comm is not None:
                    pf_comm.close()
                    pf_comm, why = None, "another rank could not create it"
            pf_collective = "engine's RCCL communicator on the batch's stream (sipnet_comm_all_gather)" if pf_comm is not None \
                else "torch.distributed process group (direct communicator failed: %s)" % why
    pf_totals = torch.ones(max(args.steps + args.warmup + 1, 1), dtype=torch.int64, device=b.device)
    pf_cycle = [0]

    def one_pass(record, plain=False):
        """plain: the pass of a single-GPU run (no exchange path at all) -- what --force-dist
        compares the distributed pass with"""
        if overlap and not plain:
            buf = bufs[npass[0] & 1]
            npass[0] += 1
            main = torch.cuda.current_stream()
            if buf["done"] is not None:
                main.wait_event(buf["done"])       # its statistics (two passes ago) have left
            b.setup()
            # the step kernel leaves the per-(step, site) sums behind (its light wave adds up the
            # plane tiles while they are in L2; sipnet_batch_run_stats) ...
            if args.gather == "sums":   # ... or every member's daily sums (no planes are written at all)
                b.run_sums(0, T, args.sum_steps, out=buf["sums"])
            else:
                b.run_stats(0, T, planes=buf["planes"], stats=buf["stats"])
            buf["ran"].record(main)
            with torch.cuda.stream(side):   # ... and the block (0.84 MB of statistics / 90 MB of c10k's daily sums) travels under the next pass
                side.wait_event(buf["ran"])
                all_gather_into(buf["gathered"], buf["sums" if args.gather == "sums" else "stats"])
                buf["done"] = torch.cuda.Event()
                buf["done"].record(side)
            return
        if pf:
            # one cycle: forecast of day k from the resampled state, analysis.  A filter does not re-run setupModel();
            # only when the forcing's days are used up does the ensemble start over (never inside the timed passes of
            # a default run).  No host round trip inside a cycle: each cycle's total weight lands in its own slot
            # and "a particle survived" is checked for all cycles after the closing barrier
            k = pf_cycle[0] % pf_days
            if k == 0 and pf_cycle[0] > 0:
                b.setup()
            # (the forecast's launch leaves the log-weights of its NEE sum too: in the analysis' own buffer, or -- a connected
            # filter -- in this rank's slice of the all-gather's)
            if distd and not plain and pf_exchange == "peer":
                sd.pf_arm_peers(b, pf_obs_k[k], pf_sigma_k[k], rank=rank, world=world, pretend_world=pretend)
            elif not (distd and not plain):
                b.pf_arm(pf_obs_k[k], pf_sigma_k[k])
            b.run(k * T, T, planes=planes)      # the time-fused step kernel
            slot = pf_totals[pf_cycle[0] % len(pf_totals):][:1]
            pf_cycle[0] += 1
            if distd and not plain and pf_exchange == "peer":
                _, info = sd.pf_analysis_peers(b, planes[0], pf_obs_k[k], pf_sigma_k[k], 0.5, rank=rank, world=world,
                                               total_out=slot, collectives=True, diagnostics=record, pretend_world=pretend, comm=pf_comm)
                if record:
                    pf_info.update(info)
                return
            _, info = sd.pf_analysis(b, planes[0], pf_obs_k[k], pf_sigma_k[k], u0=0.5, rank=rank, world=world,
                                     with_params=True, diagnostics=record, total_out=slot,
                                     collectives=distd and not plain)
            pf_info.update(info)
            return
        b.setup()                       # setupModel() for every member
        b.run(0, T, planes=planes)      # the time-fused step kernel
        if distd and not plain and args.gather != "none":
            if args.gather == "stats":
                for v in range(3):
                    b.reduce_plane(planes[v], stats[v])
                all_gather_into(gathered, stats)
            else:
                all_gather_into(gathered_full, planes)

    def barrier():
        if distd:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass(True)
    barrier()
    # (one more untimed pass behind the first synchronisation: whatever the process has deferred until then -- the
    # allocator's pending frees of the pre-passes' temporaries were seen to hold the first launch after it back by
    # 1.3 ms, 3 % of a five-pass region -- happens here, not in the timed passes)
    one_pass(False)
    barrier()
    # HIP-event kernel timing is collected in extra, untimed passes below to keep the timed
    # region free of host syncs: time K passes wall-clock first
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(False)
    barrier()
    dt = time.perf_counter() - t0
    if pf and bool((pf_totals == sa.PF_VOID_TOTAL).any()):
        raise RuntimeError("particle filter: an analysis kernel gave up at its grid barrier (not co-resident): the cycle is void")
    if pf and not bool((pf_totals > 0).all()):
        raise RuntimeError("particle filter: a cycle ended with every particle at zero weight")
    if pf:
        pi_ = b.pf_info()       # what the exchange says about itself (sipnet_batch_pf_info)
        pf_info.update({"pretend_world": pretend, "analysis_one_launch": pi_["fused"], "analysis_grid": pi_["grid"],
                        "analysis_budget": pi_["budget"], "analysis_slots": pi_["n_slots"], "params_by_index": pi_["params_by_index"],
                        "crossing_per_cycle": (pi_["crossing"] / pi_["cycles"]) if pi_["cycles"] else None})
    if distd:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else b.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # --force-dist: the same K passes without the exchange path, and the difference per pass
    dist_overhead = None
    if args.force_dist:
        pf_cycle[0] = 0
        if pf:
            b.setup()
        for _ in range(args.warmup):
            one_pass(False, plain=True)
        barrier()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            one_pass(False, plain=True)
        barrier()
        dtp = time.perf_counter() - tp0
        dist_overhead = {"ms_per_step_dist": dt / args.steps * 1e3, "ms_per_step_plain": dtp / args.steps * 1e3,
                         "dist_overhead_ms": (dt - dtp) / args.steps * 1e3,
                         "predicted_weak_scaling_efficiency": dtp / dt,
                         "backend": dist.get_backend(), "world": world,
                         "note": "K passes with the exchange path of an N-rank run (RCCL group, "
                                 + ("every member's sums over %d steps from the step kernel's own launch + all-gather of that block under the next pass's step kernel" % args.sum_steps
                                    if overlap and args.gather == "sums" else
                                    "side-stream ensemble statistics + all-gather of the statistics block under the next pass's step kernel"
                                    if overlap else ("the particle filter's ONE all-gather of log-weight blocks + peer-read resampling" if pf_exchange == "peer"
                                                     else "the particle filter's all-gather of log-weights + all-to-all of checkpoints") if pf
                                    else f"gather={args.gather} in line")
                                 + ") against K plain passes; the link time of a real N-rank exchange is not in it"}

    # dominant kernel's launch duration (HIP events on the launch stream, inside the library) and
    # the per-pass setupModel() kernel (torch events on the same, current, stream)
    kms, sms = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(max(3, min(args.steps, 5))):
        e0.record()
        b.setup()
        e1.record()
        b.time_next_launch()            # (short launches -- the filter's 48-step forecast -- are timed on request only)
        b.run(0, T, planes=planes)
        torch.cuda.synchronize()
        kms.append(b.last_kernel_ms())
        sms.append(e0.elapsed_time(e1))
    k_ms = float(np.mean(kms))
    setup_ms = float(np.mean(sms))
    li = b.last_launch()
    plan_first = li["plan_build_ms"] + li["plan_upload_ms"]   # incl. the process's first-use costs of the runtime
    # the steady-state cost of a new forcing: hand the same climate over again and re-plan
    for s in range(S):
        b.set_climate(s, clims[s])
    b.setup()
    b.run(0, T, planes=planes)
    torch.cuda.synchronize()
    li = b.last_launch()
    plan_ms = li["plan_build_ms"] + li["plan_upload_ms"]   # site plans + the record type this kernel reads
    if pf:   # the analysis step alone, torch events on the current stream (all its work is there)
        ams = []
        for _ in range(3):
            b.setup()
            b.run(0, T, planes=planes)
            barrier()
            e0.record()
            if distd and pf_exchange == "peer":
                sd.pf_analysis_peers(b, planes[0], pf_obs, pf_sigma, 0.5, rank=rank, world=world, collectives=True,
                                     pretend_world=pretend, comm=pf_comm)
            else:
                sd.pf_analysis(b, planes[0], pf_obs, pf_sigma, u0=0.5, rank=rank, world=world, with_params=True,
                               diagnostics=False, collectives=distd)
            e1.record()
            torch.cuda.synchronize()
            ams.append(e0.elapsed_time(e1))
        pf_info["analysis_ms"] = float(np.mean(ams))
        pf_info["exchange"] = pf_exchange
        pf_info["collective"] = pf_collective

    # N > 1: the north star's exchange as written -- the member-resolved NEE/GPP/ET block of every
    # rank all-gathered -- measured in an extra untimed pass: the launch is cut into 10 segments
    # and segment k's planes travel on the side stream while segment k+1 computes
    gather_full = None

    def leg_gather_full():
        nseg = 10
        cuts = [T * k // nseg for k in range(nseg + 1)]
        seglen = max(z - a for a, z in zip(cuts[:-1], cuts[1:]))
        stage = [torch.empty((world, 3, seglen, b.ncol), dtype=planes.dtype, device=b.device) for _ in range(2)]
        free = [None, None]
        barrier()
        main = torch.cuda.current_stream()
        tg0 = time.perf_counter()
        b.setup()
        for k, (a, z) in enumerate(zip(cuts[:-1], cuts[1:])):
            b.run(a, z - a, planes=planes[:, a:z])
            ran = torch.cuda.Event()
            ran.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ran)
                buf = stage[k & 1]
                seg = planes[:, a:z].contiguous() if z - a == seglen else \
                    torch.cat([planes[:, a:z], planes[:, a:a + seglen - (z - a)]], dim=1)
                all_gather_into(buf, seg)
                free[k & 1] = torch.cuda.Event()
                free[k & 1].record(side)
        barrier()
        tg = time.perf_counter() - tg0
        tmax = torch.tensor([tg], dtype=torch.float64, device="cpu" if args.rehearse else b.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return {"ms": float(tmax.item()) * 1e3, "segments": nseg,
                       "bytes_received_per_rank": int((world - 1) * 3 * T * b.ncol * planes.element_size()),
                       "note": "one pass with the member-resolved planes of every rank all-gathered "
                               "(segment k travels under the kernel of segment k+1)"}

    # (the two extra legs come after the timed region: a failure in one of them -- a node short of memory for the staging blocks --
    # is recorded in the line instead of costing the line)
    def tolerant(leg):
        try:
            return leg()
        except Exception as e:   # noqa: BLE001
            return {"error": repr(e)[:300]}

    if distd and not pf:
        gather_full = tolerant(leg_gather_full)

    # ... and the member-resolved form that fits under the kernel: every member's sums over --sum-steps steps (daily sums of a
    # half-hourly year: 1 / 48 of the planes' bytes), summed inside the step kernel's launch (sipnet_batch_run_sums) in 4 segments,
    # segment k's block travelling on the side stream while segment k+1 computes; checked against the planes of the pass above
    gather_sums = None

    def leg_gather_sums():
        K, nseg = args.sum_steps, min(4, sum_groups)
        gcuts = [sum_groups * k // nseg for k in range(nseg + 1)]
        seg_out = [torch.empty((3, z - a, b.ncol), dtype=torch.float64, device=b.device) for a, z in zip(gcuts[:-1], gcuts[1:])]
        seg_all = [torch.empty((world,) + tuple(o.shape), dtype=torch.float64, device=b.device) for o in seg_out]
        for timed_leg in (False, True):     # (once untimed: RCCL's first collective of a new size, the sums kernel's first launch)
            barrier()
            main = torch.cuda.current_stream()
            tg0 = time.perf_counter()
            b.setup()
            for k, (a, z) in enumerate(zip(gcuts[:-1], gcuts[1:])):
                b.run_sums(a * K, min(z * K, T) - a * K, K, out=seg_out[k])
                ran = torch.cuda.Event()
                ran.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ran)
                    all_gather_into(seg_all[k], seg_out[k])
            barrier()
            tg = time.perf_counter() - tg0
        tmax = torch.tensor([tg], dtype=torch.float64, device="cpu" if args.rehearse else b.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        whole = T // K * K
        ref = planes[:, :whole].double().reshape(3, whole // K, K, b.ncol).sum(2)
        mine = torch.cat([g[rank] for g in seg_all], dim=1)[:, :whole // K]
        return {"ms": float(tmax.item()) * 1e3, "segments": nseg, "sum_steps": K, "kernel": b.last_launch()["kernel"],
                       "bytes_received_per_rank": int((world - 1) * 3 * sum_groups * b.ncol * 8),
                       "bytes_sent_per_rank": int(3 * sum_groups * b.ncol * 8),
                       "max_abs_diff_vs_planes": float((mine - ref).abs().max().item()),
                       "note": "one pass with every member's sums over sum_steps steps, summed inside the step kernel's launch, "
                               "all-gathered from every rank (segment k travels under the kernel of segment k+1)"}

    if distd and not pf and b.sums_in_kernel():
        gather_sums = tolerant(leg_gather_sums)

    if args.dump_stats and not pf:
        if distd and args.gather == "stats":
            g = (bufs[(npass[0] - 1) & 1]["gathered"] if overlap else gathered)
            # ranks hold different sites in c4 and the same site's members otherwise: a rank's block
            # is added to the others' (members) or stands beside them (sites)
            total = g.sum(0) if S == 1 else torch.cat([g[r] for r in range(world)], dim=2)
        else:
            b.setup()
            b.run_stats(0, T, planes=planes, stats=stats)
            total = stats
        if rank == 0:
            np.save(args.dump_stats, total.cpu().numpy())

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
            po, _, _ = ora.run_block(flags, members[:n_chk], clims[0].slice(0, T) if pf else clims[0])
            if pf:
                pg = pf_first
            else:
                if distd:        # the planes of a whole pass from a fresh setup
                    b.setup()
                    b.run(0, T, planes=planes)
                pg = planes[:, :, :n_chk].double().cpu().numpy()
            scale = np.maximum(np.abs(po).max(axis=(1, 2), keepdims=True), 1e-3)
            flips = int(((np.abs(pg - po) / scale) > 1e-4).any(axis=(0, 1)).sum())
            parity = {"members_checked": n_chk,
                      "max_abs_dNEE": float(np.abs(pg[0] - po[0]).max()),
                      "max_abs_dGPP": float(np.abs(pg[1] - po[1]).max()),
                      "max_abs_dET": float(np.abs(pg[2] - po[2]).max()),
                      "branch_flip_members": flips,
                      "tolerance": 1e-9 if wl["prec"] == "f64" else 2e-6}
        except Exception as e:  # the checker is optional for the measurement itself
            parity = {"error": repr(e)}

    traffic, traffic_tag, valu_cycles = None, None, None
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(args.workload, {})
            # only valid for the kernel it was collected on
            if ent.get("kernel") in (None, li["kernel"]):
                traffic, traffic_tag = ent.get("hbm_bytes_per_launch"), ent.get("tag")
                valu_cycles = ent.get("valu_active_cycles_per_launch")
                # ... and for the kernel SOURCES it was collected on: a profile of an older build is quoted as such
                from sipnet_amd._lib import kernel_source_sha16
                if ent.get("source_sha16") != kernel_source_sha16():
                    traffic_tag = f"{traffic_tag} (STALE: profiled on kernel sources {ent.get('source_sha16', 'unknown')}, " \
                                  f"this tree has {kernel_source_sha16()})"
        except Exception:
            traffic = None

    # whole job, PCIe included (never `value`): hand the raw parameters and the climate over from host
    # memory, build + upload the site plan, setupModel(), the step kernel with the ensemble statistics,
    # and bring the statistics block back to the host -- what a caller pays per forcing + ensemble
    end_to_end = None
    if rank == 0 and not pf and not args.no_end_to_end:
        try:
            host_stats = torch.empty(stats.shape, dtype=torch.float64, pin_memory=True)

            def forcing(bb, pl, st, hs):
                """one forcing + ensemble handed over from host memory: climate of every site, ONE parameter upload
                for all sites (SIPNET_ALL_SITES), plan build + upload + setupModel(), the step kernel with the ensemble
                statistics from the same launch, the statistics block into pinned host memory (asynchronous)"""
                bb.set_climates(clims)          # (one call: the copies into the pinned blocks run on the plan threads)
                bb.set_params(None, members)
                bb.setup()
                bb.run_stats(0, T, planes=pl, stats=st)
                hs.copy_(st, non_blocking=True)

            e2e = []
            for _ in range(8):   # serial: the latency of one forcing
                torch.cuda.synchronize()
                te0 = time.perf_counter()
                forcing(b, planes, stats, host_stats)
                torch.cuda.synchronize()
                e2e.append(time.perf_counter() - te0)
            e2e_s = float(np.median(e2e[1:]))
            e2e_samples_ms = [round(x * 1e3, 3) for x in e2e]
            plan_device_sites = int(b.last_launch()["plan_device_sites"])   # (an idle batch: the device builds the plans, plan_device.h)
            # pipelined: forcings alternate between TWO batches on two streams, so the host side of forcing k + 1
            # (climate copies, plan build, uploads) runs under the step kernel of forcing k; uploads wait for their own
            # batch's last launch only.  Steady-state time per forcing: the median of three legs of 8 forcings (2 of warm-up before the first).
            pipelined = None
            try:
                # a caller who pipelines has made the GPU the bottleneck and has idle cores: it asks for host-built plans
                # (the default would pick them too whenever a batch's previous launch is still running -- explicit here,
                # so that the leg does not depend on which side of that race a hand-over falls)
                b2 = sa.Batch(flags, S, M, prec, device=local_rank, fast_math=bool(args.fast_math) if prec == sa.F64 else None,
                              kernel_options=sa.KOPT_HOST_PLAN)
                b.set_kernel(sa.KERNEL_AUTO, sa.KOPT_HOST_PLAN)
                planes_b, _ = b2.alloc_outputs(T)
                lanes = [dict(b=b, pl=planes, st=stats, hs=host_stats, s=torch.cuda.Stream(device=b.device)),
                         dict(b=b2, pl=planes_b, st=torch.empty_like(stats), hs=torch.empty_like(host_stats).pin_memory(),
                              s=torch.cuda.Stream(device=b.device))]
                torch.cuda.synchronize()
                legs = []
                for leg in range(3):           # three legs of 8 forcings after 2 of warm-up: the median leg (one stall of the
                    tp0 = None                 # box -- a page fault storm, another tenant -- moves a leg's mean by milliseconds)
                    for k in range(10 if leg == 0 else 8):
                        if k == (2 if leg == 0 else 0):
                            torch.cuda.synchronize()
                            tp0 = time.perf_counter()
                        ln = lanes[k & 1]
                        with torch.cuda.stream(ln["s"]):
                            forcing(ln["b"], ln["pl"], ln["st"], ln["hs"])
                    torch.cuda.synchronize()
                    legs.append((time.perf_counter() - tp0) / 8)
                pipelined = float(np.median(legs))
                b.set_kernel(sa.KERNEL_AUTO, 0)
                b2.close()
                del planes_b
            except Exception as e:
                pipelined = repr(e)
            end_to_end = {"ms": e2e_s * 1e3, "ms_samples": e2e_samples_ms, "value": per_launch_units / e2e_s, "unit": "ensemble-site-timesteps/s",
                          "pipelined_ms": pipelined * 1e3 if isinstance(pipelined, float) else None,
                          "pipelined_ms_legs": [round(x * 1e3, 3) for x in legs] if isinstance(pipelined, float) else None,
                          "pipelined_value": per_launch_units / pipelined if isinstance(pipelined, float) else None,
                          **({"pipelined_error": pipelined} if isinstance(pipelined, str) else {}),
                          "bytes_up": int(members.nbytes + S * (clims[0].data.nbytes + clims[0].year.nbytes + clims[0].day.nbytes)),
                          "bytes_down": int(host_stats.numel() * 8),
                          "plan_device_sites": plan_device_sites,   # sites whose plan the device built in the serial leg (the pipelined
                                                                    # leg hands forcings to busy batches: the host's cores build)
                          "includes": "climate of every site (one call) + raw parameters (one upload for all sites) from host memory, "
                                      "the site plans (built on the device from the climate where eligible, else host-built + uploaded), setupModel(), the step kernel with the ensemble statistics from "
                                      "the same launch (sipnet_batch_run_stats), the statistics block into pinned host memory; "
                                      "ms: one forcing, nothing overlapped (median of 7 after one warm-up; ms_samples: all 8); pipelined_ms: per "
                                      "forcing with two batches in flight (the host side of forcing k + 1 under the kernel of "
                                      "forcing k; these batches ask for host-built plans, SIPNET_KOPT_HOST_PLAN: the GPU is the "
                                      "bottleneck there and the cores are idle); the member-resolved planes stay in HBM"}
        except Exception as e:
            end_to_end = {"error": repr(e)}

    probe = None
    if rank == 0 and world == 1 and not args.no_fill_probe and not pf and args.fast_math:
        try:
            probe = fill_probe(sa, synth, flags, base, wl["prec"], local_rank)
        except Exception as e:
            probe = {"error": repr(e)}

    if rank == 0:
        waves_per_block = li["block_threads"] // 64
        if "Pair" in li["kernel"]:
            waves_per_block = 6      # two of the eight wavefronts only keep the SIMD rotation and leave at once
        cus_used = min(li["grid"], li["num_cus"]) if "Coop" in li["kernel"] else min((li["grid"] + 3) // 4, li["num_cus"])
        simds_used = min(li["grid"] * min(waves_per_block, 4), 4 * li["num_cus"])
        line = {
            "metric": "ensemble-site-timesteps/sec", "value": value,
            "unit": "ensemble-site-timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl["prec"], "data": "synthetic",
            "config": {"workload": f"{args.workload}: {S} site(s) x {M} members per GPU x {T} "
                                   f"half-hourly steps, perturbed params, " + ("default flags" if not wl.get("flags") else "flags " + "+".join(sorted(wl["flags"])))
                                   + (", particle-filter cycle (forecast of day k from the resampled state + analysis; no setupModel() in a cycle)" if pf else ""),
                       "sites_per_gpu": S, "members_per_site": M, "timesteps": T,
                       "fast_math": bool(args.fast_math),
                       "gather": args.gather if distd else "n/a (1 GPU)",
                       **({"dist_overhead": dist_overhead} if dist_overhead else {}),
                       "parallelism": f"ensemble-sharded x{world}",
                       "ranks_seen": ranks_seen, "devices_seen": devices_seen, "device_ids": device_ids,
                       **({"gather_full": gather_full} if gather_full else {}),
                       **({"gather_sums": gather_sums} if gather_sums else {}),
                       **({"particle_filter": pf_info} if pf else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_tag": traffic_tag,
                         "hbm_measured_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "kernel": li["kernel"], "kernel_ms": k_ms,
                         "grid": li["grid"], "block_threads": li["block_threads"],
                         "waves_per_simd": li["grid"] * waves_per_block / (4.0 * li["num_cus"]),
                         "waves_per_simd_register_budget": li["waves_per_simd"],
                         "cus_used": cus_used, "cus_total": li["num_cus"],
                         "simds_used": simds_used, "simds_total": 4 * li["num_cus"],
                         "lds_bytes_per_workgroup": li["lds_bytes"],
                         # the PHYSICAL ceiling beside the contract number (SURVEY 8(d) "ALU cross-check"): the share of
                         # SIMD-cycles in which a vector ALU was executing -- SQ_ACTIVE_INST_VALU x 4 of the committed
                         # profile (profiles/pmc_traffic.json, like `traffic`) over SIMDs x this run's kernel time at
                         # the 2.4 GHz peak clock (a lower clock under load makes the true fraction higher)
                         "valu_busy_frac": (valu_cycles / (4.0 * li["num_cus"] * k_ms * 1e-3 * SCLK_PEAK_HZ)) if valu_cycles else None,
                         "valu_busy_frac_of_used_simds": (valu_cycles / (simds_used * k_ms * 1e-3 * SCLK_PEAK_HZ)) if valu_cycles else None,
                         "valu_clock_ghz_assumed": SCLK_PEAK_HZ / 1e9,
                         "issue_frac": (per_launch_units / (k_ms * 1e-3) / probe["rate"]) if probe and "rate" in probe else None,
                         "fill_probe": probe,
                         "plan_ms": plan_ms, "plan_ms_first_in_process": plan_first, "plan_build_ms": li["plan_build_ms"], "plan_upload_ms": li["plan_upload_ms"], "plan_threads": li["plan_threads"], "setup_ms": setup_ms,
                         "end_to_end": end_to_end,
                         "algorithmic_bytes_per_unit": ALGO_BYTES[wl["prec"]],
                         "units_per_launch": per_launch_units},
            "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line))
    if distd:
        dist.destroy_process_group()
    b.close()


if __name__ == "__main__":
    main()
