#!/usr/bin/env python3
"""dev (GPU box): step-kernel time of a batch with more than four chunks per CU on the kernels that can take it
(one-wave = AUTO's choice there, and the four-chunk cooperative kernel with its workgroups queued behind each
other).  usage: big_batch_time.py [members=131072] [f32|f64] [steps=17520]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import sipnet_amd as sa
from sipnet_amd import synth

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
prec = sa.F64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else sa.F32_MIXED
T = int(sys.argv[3]) if len(sys.argv) > 3 else 17520
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
members = synth.perturbed_params(base, M)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
for name, kern in (("auto", sa.KERNEL_AUTO), ("coop_quad", sa.KERNEL_COOP_QUAD), ("coop_pair", sa.KERNEL_COOP_PAIR)):
    b = sa.Batch(flags, 1, M, prec, fast_math=True if prec == sa.F64 else None, kernel=kern)
    b.set_climate(0, clim); b.set_params(0, members)
    planes, _ = b.alloc_outputs(T)
    ms = []
    for _ in range(3):
        b.setup(); b.run(0, T, planes=planes); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
    print("%-10s %s  %.3f ms  = %.1f G steps/s" % (name, b.last_launch()["kernel"], min(ms), M * T / min(ms) / 1e6))
    b.close(); del planes
    torch.cuda.empty_cache()
