#!/usr/bin/env python3
"""dev check: cooperative kernel vs one-wave kernel, bitwise, growing sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
os.environ["SIPNET_FAST_MATH"] = "1"
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
def run(M, T, coop, prec=sa.F64):
    os.environ["SIPNET_COOP"] = str(coop)
    b = sa.Batch(flags, 1, M, prec)
    b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
    b.set_params(0, synth.perturbed_params(base, M))
    b.setup(); t0 = time.time(); planes, _ = b.run(); torch.cuda.synchronize(); dt = time.time() - t0
    p = planes.cpu().numpy(); st = b.get_state(); ms = b.last_kernel_ms(); b.close()
    return p, st, ms
for M, T in [(64, 50), (64, 2000), (200, 17520), (10240, 17520)]:
    for prec in (sa.F64, sa.F32_MIXED):
        p1, s1, ms1 = run(M, T, 1, prec); print("coop ran", M, T, prec, ms1, flush=True)
        p0, s0, ms0 = run(M, T, 0, prec)
        print(f"M={M} T={T} prec={prec}: coop {ms1:.3f} ms, single {ms0:.3f} ms, planes equal {np.array_equal(p0, p1)}, "
              f"state equal {np.array_equal(s0[:, :29], s1[:, :29])}, max|d| {np.abs(p0.astype(np.float64) - p1).max():.3e}", flush=True)
