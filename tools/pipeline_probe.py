#!/usr/bin/env python3
"""dev: the pipelined end-to-end leg of bench.py, call by call (where does the host block?)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
if os.environ.get('SIPNET_LIB'):
    from sipnet_amd import _lib
    _lib.use_library(os.environ['SIPNET_LIB'])
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
lanes = []
for k in range(2):
    b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None)
    planes, _ = b.alloc_outputs(T)
    st = torch.empty((3, T, S, 2), dtype=torch.float64, device=b.device)
    lanes.append(dict(b=b, pl=planes, st=st, hs=torch.empty(st.shape, dtype=torch.float64, pin_memory=True), s=(torch.cuda.ExternalStream(sa.lib().sipnet_stream_create(0)) if os.environ.get("OWN_STREAMS") else torch.cuda.Stream())))
torch.cuda.synchronize()
t00 = time.perf_counter()
for k in range(8):
    ln = lanes[k & 1]
    bb = ln["b"]
    with torch.cuda.stream(ln["s"]):
        t = [time.perf_counter()]
        bb.set_climates(clims)
        t.append(time.perf_counter())
        bb.set_params(None, members); t.append(time.perf_counter())
        bb.setup(); t.append(time.perf_counter())
        bb.run_stats(0, T, planes=ln["pl"], stats=ln["st"]); t.append(time.perf_counter())
        ln["hs"].copy_(ln["st"], non_blocking=True); t.append(time.perf_counter())
    print(k, "at %.2f:" % (1e3 * (t[0] - t00)), " ".join("%.2f" % (1e3 * (b_ - a_)) for a_, b_ in zip(t[:-1], t[1:])), flush=True)
torch.cuda.synchronize()
print("total per forcing %.2f ms" % (1e3 * (time.perf_counter() - t00) / 8))
