#!/usr/bin/env python3
"""What the in-kernel ensemble statistics cost (dev tool, GPU): step kernel time of run() against
run_stats() (HIP events of the library around the step kernel), and the whole call incl. the
second-stage kernel (torch events), per workload.  usage: stats_cost.py c10k c4 c3"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sipnet_amd import _lib
if os.environ.get("SIPNET_VARIANT"):
    _lib.use_library(os.path.join(REPO, "build", "variants", os.environ["SIPNET_VARIANT"], "libsipnet_amd.so"))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from bench import WORKLOADS

for name in sys.argv[1:] or ["c10k"]:
    wl = WORKLOADS[name]
    flags = sa.flags_from(**wl.get("flags", {}))
    base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", wl.get("param", "base_forest.param")), flags)
    S, M, T = wl["sites"], wl["members"], wl["steps"]
    prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
    b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None)
    members = synth.perturbed_params(base, M)
    for s in range(S):
        raw = synth.half_hourly_year_raw(T, site=s)
        light = os.environ.get("LIGHT", "asis")      # night / day: polar night / midnight sun
        if light == "night":
            raw["par"][:] = 0.0
        if light == "day":
            raw["par"] = np.maximum(raw["par"], 2.0)
        b.set_climate(s, synth.convert_raw(synth.round_like_file(raw)))
        b.set_params(s, members)
    planes, _ = b.alloc_outputs(T)
    stats = torch.empty((3, T, S, 2), dtype=torch.float64, device=planes.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for mode in os.environ.get("MODES", "run,run_stats,run,run_stats").split(","):
        k, w = [], []
        for _ in range(4):
            b.setup()
            torch.cuda.synchronize()
            e0.record()
            if mode == "run":
                b.run(0, T, planes=planes)
            else:
                b.run_stats(0, T, planes=planes, stats=stats)
            e1.record()
            torch.cuda.synchronize()
            k.append(b.last_kernel_ms()); w.append(e0.elapsed_time(e1))
        res.setdefault(mode, []).append((min(k), min(w)))
    print(os.environ.get("SIPNET_VARIANT", "product"), os.environ.get("LIGHT", "asis"), name, b.last_launch()["kernel"], {m: ["kernel %.3f call %.3f ms" % x for x in v] for m, v in res.items()})
    b.close()
