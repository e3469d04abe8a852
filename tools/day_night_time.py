#!/usr/bin/env python3
"""dev (GPU box): c10k with the synthetic year's light forced to polar night / midnight sun, to
read off what a night step and a day step of the cooperative kernel cost (the day step carries the
leaf-area -> potential-photosynthesis -> photosynthesis chain through all three wavefronts).
usage: [SIPNET_LIB=build/variants/<name>/libsipnet_amd.so] [M=members] [PREC=f64|f32] day_night_time.py [kernel: auto|coop_lds|coop_hbm|coop_pair|coop_quad|one_wave]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if os.environ.get("SIPNET_LIB"):                 # an experimental build from tools/build_variants.py
    from sipnet_amd import _lib
    _lib.use_library(os.environ["SIPNET_LIB"])
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth

kern = dict(auto=sa.KERNEL_AUTO, coop_lds=sa.KERNEL_COOP_LDS, coop_hbm=sa.KERNEL_COOP_HBM, coop_pair=sa.KERNEL_COOP_PAIR,
            coop_quad=sa.KERNEL_COOP_QUAD, one_wave=sa.KERNEL_ONE_WAVE)[sys.argv[1] if len(sys.argv) > 1 else "auto"]
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
M, T = int(os.environ.get("M", "10240")), 17520
members = synth.perturbed_params(base, M)
for name in ("as is", "night", "day"):
    raw = synth.half_hourly_year_raw(T)
    if name == "night": raw["par"][:] = 0.0
    if name == "day": raw["par"] = np.maximum(raw["par"], 2.0)
    clim = synth.convert_raw(synth.round_like_file(raw))
    f32 = os.environ.get("PREC", "f64") == "f32"
    b = sa.Batch(flags, 1, M, sa.F32_MIXED if f32 else sa.F64, fast_math=None if f32 else True, kernel=kern)
    b.set_climate(0, clim); b.set_params(0, members)
    planes, _ = b.alloc_outputs(T)
    ms = []
    for r in range(4):
        b.setup(); b.run(0, T, planes=planes); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
    day_share = float((raw["par"] > 0).mean())
    print("%-6s day steps %.3f  %s  %.3f ms  = %.0f cycles/step at 2.4 GHz" %
          (name, day_share, b.last_launch()["kernel"], min(ms), min(ms) * 1e-3 * 2.4e9 / T))
    b.close()
