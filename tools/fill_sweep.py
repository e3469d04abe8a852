#!/usr/bin/env python3
"""dev: throughput of the one-wave kernel as the chip fills up (1 site, fp32-mixed / fp64,
default flags, synthetic year); outputs only NEE to keep the planes small"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
T = 17520
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
for prec, name in ((sa.F32_MIXED, "f32-mixed"), (sa.F64, "f64")):
    for M in (65536, 131072, 262144, 524288):
        if prec == sa.F64 and M > 262144:
            continue
        b = sa.Batch(flags, 1, M, prec, fast_math=True)
        b.set_climate(0, clim); b.set_params(0, synth.perturbed_params(base, M))
        dt = torch.float32 if prec == sa.F32_MIXED else torch.float64
        plane = torch.empty((T, M), dtype=dt, device="cuda")
        ptr = lambda x: x.data_ptr()
        import ctypes as C
        ms = []
        for r in range(2):
            b.setup()
            sa.lib().sipnet_batch_run(b.h, 0, T, C.c_void_p(ptr(plane)), None, None, None, M, b._stream())
            torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
        b.close(); del plane; torch.cuda.empty_cache()
        waves = (M + 63) // 64
        print(f"{name:10s} {M:7d} members ({waves/1024:.1f} waves/SIMD): {min(ms):8.2f} ms  {M*T/min(ms)/1e6:7.1f} G steps/s", flush=True)
