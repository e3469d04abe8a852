#!/usr/bin/env python3
"""dev: throughput of the strict all-flags kernel (what non-default flag sets run on)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
T, M = 17520, 10240
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
for name, kw, fm in (("default flags, fast math (throughput kernels)", {}, "1"),
                     ("default flags, strict", {}, "0"),
                     ("litterPool+N+anaerobic, fast math", dict(litterPool=1, nitrogenCycle=1, anaerobic=1), "1"),
                     ("litterPool+N+anaerobic, strict", dict(litterPool=1, nitrogenCycle=1, anaerobic=1), "0")):
    os.environ["SIPNET_FAST_MATH"] = fm
    flags = sa.flags_from(**kw)
    base, _ = sa.read_params("tests/golden/synth/allflags.param", flags)
    b = sa.Batch(flags, 1, M, sa.F64); b.set_climate(0, clim); b.set_params(0, synth.perturbed_params(base, M))
    b.setup(); b.run(); torch.cuda.synchronize(); b.setup(); b.run(); ms = b.last_kernel_ms(); b.close()
    print(f"{name:50s}: {ms:8.2f} ms  {M*T/ms/1e6:7.2f} G steps/s", flush=True)
