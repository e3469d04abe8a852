#!/usr/bin/env python3
"""dev (GPU box): the nitrogen-cycle kernels with the record / every accumulator (SIPNET_KOPT_FULL_STATE) and with the diagnostics
counters at c10k's (one chunk per CU) and c4's (two) shapes: kernel ms per year, per library (SIPNET_LIB) or the product.
usage: [SIPNET_LIB=...] ncyc_full_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
T = 17520
flags = sa.flags_from(litterPool=1, anaerobic=1, nitrogenCycle=1)
base = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "allflags_forest.param"), flags)[0]
for S, M in ((1, 10240), (32, 1024)):
    for what in ("lean", "full_state", "diagnostics", "diagnostics, one-wave kernel forced"):
        b = sa.Batch(flags, S, M, sa.F64, fast_math=True, kernel_options=sa.KOPT_FULL_STATE if what == "full_state" else 0,
                     kernel=sa.KERNEL_ONE_WAVE if "one-wave" in what else sa.KERNEL_AUTO)
        for s in range(S):
            b.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
        b.set_params(None, synth.perturbed_params(base, M))
        if what.startswith("diagnostics"):
            b.enable_diagnostics()
        planes, _ = b.alloc_outputs(T)
        ms = []
        for _ in range(4):
            b.setup()
            try:
                b.run(0, T, planes=planes)
            except Exception as e:
                print(what, S, M, "refused:", str(e)[:80]); break
            torch.cuda.synchronize()
            ms.append(b.last_kernel_ms())
        if ms:
            print("%-12s %2d x %5d members, %-36s %8.3f ms  %s" % (os.environ.get("SIPNET_LIB", "product")[-40:], S, M, what, min(ms[1:]), b.last_launch()["kernel"]), flush=True)
        b.close()
