#!/usr/bin/env python3
"""Profiling target: one setup + run (+ plane reductions as a traffic calibration
kernel with an exactly known byte count) of a bench workload.  Used under rocprofv3."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import sipnet_amd as sa
from sipnet_amd import synth
from bench import WORKLOADS

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c10k"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
flags = sa.flags_from(**wl.get("flags", {}))
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", wl.get("param", "base_forest.param")), flags)
S, M, T = wl["sites"], wl["members"], wl["steps"]
prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None)
members = synth.perturbed_params(base, M)
for s in range(S):
    b.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
    b.set_params(s, members)
planes, _ = b.alloc_outputs(T)
for _ in range(reps):
    b.setup()
    b.run(0, T, planes=planes)
    st = b.reduce_plane(planes[0])
torch.cuda.synchronize()
print("kernel ms", b.last_kernel_ms(), "plane bytes", planes[0].numel() * planes[0].element_size())
print("step kernel name", b.last_launch()["kernel"])
from sipnet_amd._lib import kernel_source_sha16
print("kernel sources sha16", kernel_source_sha16())
