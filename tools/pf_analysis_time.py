#!/usr/bin/env python3
"""dev (GPU box): the one-batch particle-filter analysis (sipnet_batch_pf_analysis) alone, at c5's shape: ms per call
over K calls (HIP events), per library under build/variants (SIPNET_LIB) or the product.
usage: [SIPNET_LIB=...] pf_analysis_time.py [n_particles] [calls]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T = 48
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
b = sa.Batch(flags, 1, n, sa.F32_MIXED)
b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
b.set_params(0, synth.perturbed_params(base, n))
b.setup()
planes, _ = b.run(0, T)
tot = planes[0].double().sum(0)
obs, sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
total = torch.zeros(1, dtype=torch.int64, device=b.device)
for with_params in (True, False):
    for _ in range(5):
        b.pf_analysis_local(planes[0], obs, sigma, 0.5, with_params, total)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K):
        b.pf_analysis_local(planes[0], obs, sigma, 0.5, with_params, total)
    e1.record()
    torch.cuda.synchronize()
    print("%s: n=%d with_params=%d  %.4f ms per analysis (fused weights/scan/ancestors + gather), total weight %d" %
          (os.environ.get("SIPNET_LIB", "product"), n, with_params, e0.elapsed_time(e1) / K, int(total.item())))

L = sa.lib()
if hasattr(L, "sipnet_debug_read_pf_stamps"):      # a -DSIPNET_PF_STAMPS build: workgroup 0's clock (100 MHz) at every phase
    import ctypes as C
    st = (C.c_ulonglong * 8)()
    L.sipnet_debug_read_pf_stamps(st)
    t = [int(x) for x in st]
    print("phases of the last launch, workgroup 0 (us): " + ", ".join("%s %.2f" % (nm, (t[j] - t[i]) / 100.0) for nm, i, j in (
        ("log-weights + chunk maximum", 0, 1), ("barrier 1", 1, 2), ("weights + block scan", 2, 3), ("barrier 2", 3, 4),
        ("chunk offsets", 4, 5), ("ancestors", 5, 6))))
