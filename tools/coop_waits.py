#!/usr/bin/env python3
"""dev (GPU box): who waits for whom in the cooperative kernel.  Needs the diagnostic build
`python tools/build_variants.py waits=-DSIPNET_WAITS` (sums s_memtime differences around every
hand-over wait of workgroup 0).  Prints cycles per step (100 MHz s_memtime ticks x 24 at 2.4 GHz)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sipnet_amd import _lib
_lib.use_library(os.path.join(REPO, "build", "variants", "waits", "libsipnet_amd.so"))
import torch, sipnet_amd as sa
from sipnet_amd import synth
ncyc = bool(os.environ.get("NCYC"))          # NCYC=1: the nitrogen-cycle flag set (stepCoopNKernel)
flags = sa.flags_from(litterPool=1, anaerobic=1, nitrogenCycle=1) if ncyc else sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "allflags_forest.param" if ncyc else "base_forest.param"), flags)
M, T = 10240, 17520
b = sa.Batch(flags, 1, M, sa.F64, fast_math=True, kernel_options=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
raw = synth.half_hourly_year_raw(T)
light = os.environ.get("LIGHT", "asis")        # day / night: midnight sun / polar night (see day_night_time.py)
if light == "night": raw["par"][:] = 0.0
if light == "day":
    import numpy as np
    raw["par"] = np.maximum(raw["par"], 2.0)
b.set_climate(0, synth.convert_raw(synth.round_like_file(raw)))
b.set_params(0, synth.perturbed_params(base, M))
runner = b.run_stats if os.environ.get("STATS") else (lambda: b.run(want_planes=True))   # STATS=1: with the light wave's statistics
b.setup(); runner(); torch.cuda.synchronize()
b.setup(); runner(); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
_lib.lib().sipnet_debug_read_coop_waits(out)
tick = 1.0    # s_memtime ticks are core-clock cycles on this part (total = kernel time x 2.4 GHz)
names = {0: "L: wait for lai", 2: "L: statistics loads (vmcnt)", 1: "L: wait for C before posting factors", 3: "L: total", 4: "W: take pgp+alive", 5: "W: wait for C's progress", 7: "W: total",
         8: "C: take factors + moisture (+record)", 9: "C: take psn", 10: "C: late take of S's mineral N (NCYC)", 11: "C: total",
         12: "S: take moisture terms (NCYC)", 13: "S: take plant fluxes", 14: "S: take leached share", 15: "S: total"}
print("light:", light, "kernel", b.last_launch()["kernel"], "%.2f ms" % b.last_kernel_ms())
for k in sorted(names):
    print("%-28s %8.0f cycles/step" % (names[k], out[k] * tick / T))
