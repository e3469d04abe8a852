#!/usr/bin/env python3
"""dev: throughput with optional model flags on (fast math: step_fast.hip run-time-flag /
N-cycle instantiations; strict: stepKernel).  usage: flags_throughput.py [members] [f64|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
T = 17520
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
PREC = sa.F32_MIXED if len(sys.argv) > 2 and sys.argv[2] == 'f32' else sa.F64
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
for name, kw, fm in (("default flags, fast math (throughput kernels)", {}, "1"),
                     ("default flags, fast math, one-wave kernel", {}, "ow"),
                     ("default flags, one-wave kernel, run-time flags", {}, "owrt"),
                     ("default flags, strict", {}, "0"),
                     ("litterPool+N+anaerobic, fast math", dict(litterPool=1, nitrogenCycle=1, anaerobic=1), "1"),
                     ("litterPool+N+anaerobic, strict", dict(litterPool=1, nitrogenCycle=1, anaerobic=1), "0"),
                     ("litterPool+N+anaerobic, run-time flags", dict(litterPool=1, nitrogenCycle=1, anaerobic=1), "rt"),
                     ("all optional flags on, fast math", dict(litterPool=1, nitrogenCycle=1, anaerobic=1, carbonSaturation=1, flooding=1, growthResp=1, leafWater=1), "1"),
                     ("litterPool only, fast math", dict(litterPool=1), "1")):
    if PREC != sa.F64 and fm == "0":
        continue
    kern, kopt = sa.KERNEL_AUTO, 0
    if fm in ("ow", "owrt"):
        kern = sa.KERNEL_ONE_WAVE
        if fm == "owrt":
            kopt = sa.KOPT_RUNTIME_FLAGS
        fm = "1"
    if fm == "rt":
        kopt = sa.KOPT_RUNTIME_FLAGS
        fm = "1"
    flags = sa.flags_from(**kw)
    base, _ = sa.read_params("tests/golden/synth/allflags.param", flags)
    b = sa.Batch(flags, 1, M, PREC, fast_math=(fm == "1") if PREC == sa.F64 else None, kernel=kern, kernel_options=kopt); b.set_climate(0, clim); b.set_params(0, synth.perturbed_params(base, M))
    b.setup(); b.run(); torch.cuda.synchronize(); b.setup(); b.run(); ms = b.last_kernel_ms(); b.close()
    print(f"{name:50s}: {ms:8.2f} ms  {M*T/ms/1e6:7.2f} G steps/s", flush=True)
