#!/usr/bin/env python3
"""dev: where the whole-job time of a multi-site batch goes (bench.py's end_to_end leg, phase by phase).
usage: e2e_breakdown.py [workload] [host]     (host: every site plan on host threads, SIPNET_KOPT_HOST_PLAN)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from bench import WORKLOADS
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
flags = sa.flags_from(**wl.get("flags", {}))
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", wl.get("param", "base_forest.param")), flags)
S, M, T = wl["sites"], wl["members"], wl["steps"]
prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(S)]
members = synth.perturbed_params(base, M)
HOST = len(sys.argv) > 2 and sys.argv[2] == 'host'
b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None, kernel_options=sa.KOPT_HOST_PLAN if HOST else 0)
planes, _ = b.alloc_outputs(T)
stats = torch.empty((3, T, S, 2), dtype=torch.float64, device=b.device)
host = torch.empty(stats.shape, dtype=torch.float64, pin_memory=True)
def sync(): torch.cuda.synchronize()
for rep in range(4):
    t = [time.perf_counter()]
    def mark():
        sync(); t.append(time.perf_counter())
    b.set_climates(clims)
    mark()
    b.set_params(None, members)
    mark()
    b.setup()
    mark()
    b.run_stats(0, T, planes=planes, stats=stats)
    mark()
    host.copy_(stats, non_blocking=True)
    mark()
    x = stats.cpu()
    mark()
    names = ["set_climate", "set_params(ALL_SITES)", "setup (plan build + upload + setupModel)", "run_stats", "stats -> pinned host", "stats -> pageable host"]
    if rep:
        print("  ".join(f"{n} {1e3 * (b_ - a_):.2f}" for n, a_, b_ in zip(names, t[:-1], t[1:])), f"| total {1e3 * (t[-2] - t[0]):.2f} ms (pinned)", flush=True)
# the same hand-over without a synchronisation between the phases (what a caller does)
for rep in range(4):
    sync(); t0 = time.perf_counter()
    b.set_climates(clims)
    b.set_params(None, members)
    b.setup()
    t1 = time.perf_counter()
    b.run_stats(0, T, planes=planes, stats=stats)
    host.copy_(stats, non_blocking=True)
    sync(); t2 = time.perf_counter()
    if rep:
        print(f"one forcing alone, no synchronisation inside: host side until the step kernel is queued {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms", flush=True)
li = b.last_launch()
print("plan built by", "the host" if HOST else "the device where eligible", "-- device-built sites:", li["plan_device_sites"])
print("plan_build_ms", li["plan_build_ms"], "plan_upload_ms", li["plan_upload_ms"], "kernel", li["kernel"], b.last_kernel_ms())
