#!/usr/bin/env python3
"""dev (GPU box): c10k kernel time with the 44-column record -- strict kernel (round-1 path of the
ensemble CLI) vs the Full instantiations of the throughput kernels.
usage: full_record_time.py [members] [default|russell_2|russell_3]   (the flag set; russell_2 = litter pool + anaerobic
+ nitrogen cycle, russell_3 = growth respiration + leaf water + litter pool, no moisture effect)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import sipnet_amd as sa
from sipnet_amd import synth
SETS = {"default": ({}, "base_forest.param"), "russell_2": (dict(litterPool=1, anaerobic=1, nitrogenCycle=1), "allflags_forest.param"),
        "russell_3": (dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0), "allflags_forest.param")}
which = sys.argv[2] if len(sys.argv) > 2 else "default"
flags = sa.flags_from(**SETS[which][0])
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", SETS[which][1]), flags)
print("flag set:", which)
M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 10240, 17520
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
members = synth.perturbed_params(base, M)
rec = torch.empty((T, sa.NREC, M), dtype=torch.float64, device="cuda")
for name, fast, kern in (("strict-order kernel, strict math", False, sa.KERNEL_AUTO),
                         ("strict-order kernel, fast math", True, sa.KERNEL_STRICT),
                         ("cooperative Full (auto)", True, sa.KERNEL_AUTO),
                         ("one-wave Full", True, sa.KERNEL_ONE_WAVE)):
    b = sa.Batch(flags, 1, M, sa.F64, fast_math=fast, kernel=kern)
    b.set_climate(0, clim); b.set_params(0, members)
    ms = []
    for _ in range(3):
        b.setup(); b.run(0, T, rec=rec, want_planes=False); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
    print(f"{name:36s} {b.last_launch()['kernel']:48s} {min(ms):8.2f} ms  {M*T/min(ms)/1e6:6.2f} G steps/s", flush=True)
    b.close()
