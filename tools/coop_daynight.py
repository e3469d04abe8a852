#!/usr/bin/env python3
"""dev: cooperative kernel step time for all-night / normal / all-day forcing"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
os.environ["SIPNET_FAST_MATH"] = "1"
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
T, M = 17520, 10240
clim0 = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
for coop in (1, 0):
    os.environ["SIPNET_COOP"] = str(coop)
    for name, f in (("normal", None), ("all night", 0.0), ("all day", 1.0)):
        d = clim0.data.copy()
        if f == 0.0: d[:, 3] = 0.0
        if f == 1.0: d[:, 3] = np.maximum(d[:, 3], 5.0)
        c = sa.ClimTable(d, clim0.year, clim0.day)
        b = sa.Batch(flags, 1, M, sa.F64); b.set_climate(0, c); b.set_params(0, synth.perturbed_params(base, M))
        b.setup(); b.run(); torch.cuda.synchronize(); b.setup(); b.run(); ms = b.last_kernel_ms(); b.close()
        print(f"coop={coop} {name:10s}: {ms:.3f} ms  {ms*1e-3*2.4e9/T:.0f} cycles/step", flush=True)
