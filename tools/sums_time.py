#!/usr/bin/env python3
"""dev (GPU box): the in-kernel sums (sipnet_batch_run_sums, daily groups) against the planes at a workload's shape: kernel ms
(the library's HIP events), per library under build/variants (SIPNET_LIB) or the product.
usage: [SIPNET_LIB=...] sums_time.py [members=10240] [sites=1] [reps=5] [f64|f32]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
prec = sa.F32_MIXED if len(sys.argv) > 4 and sys.argv[4] == "f32" else sa.F64
T = 17520
flags = sa.flags_from()
base = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)[0]
b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None)
for s in range(S):
    b.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
b.set_params(None, synth.perturbed_params(base, M))
planes, _ = b.alloc_outputs(T)
sums = torch.empty((3, T // 48, b.ncol), dtype=torch.float64, device=b.device)
res = {}
for what in ("planes", "sums"):
    ms = []
    for _ in range(reps + 1):
        b.setup()
        if what == "planes":
            b.run(0, T, planes=planes)
        else:
            b.run_sums(0, T, 48, out=sums)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    res[what] = (min(ms[1:]), float(np.median(ms[1:])), b.last_launch()["kernel"])
ref = planes.double().view(3, T // 48, 48, -1).sum(2)
err = float((sums - ref).abs().max())
print("%s: %s %d x %d members: planes %.4f ms (%s) | sums %.4f ms min, %.4f med (%s) | max|sums - summed planes| %.2e" % (
    os.environ.get("SIPNET_LIB", "product")[-40:], "fp64" if prec == sa.F64 else "fp32-mixed", S, M, res["planes"][0], res["planes"][2], res["sums"][0], res["sums"][1], res["sums"][2], err))
