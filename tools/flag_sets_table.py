#!/usr/bin/env python3
"""Measured table of every optional flag set (tests/test_gpu_flags.py's 13 + russell_3's) on the kernel
SIPNET_KERNEL_AUTO picks, at c10k's shape (1 site x 10 240 members) and c4's (32 sites x 1 024), fp64 fast
math, one synthetic half-hourly year.  usage: flag_sets_table.py [out.md] [shape ...]
shapes: c10k c4 (fp64) | c3f32 = 1 site x 65 536 members fp32-mixed (four chunks per CU) | c10krec = c10k's shape with
the 44-column record (the Full builds)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import sipnet_amd as sa
from sipnet_amd import synth

SETS = {
    "default": dict(),
    "growth_resp": dict(growthResp=1),
    "leaf_water": dict(leafWater=1),
    "litter_pool": dict(litterPool=1),
    "no_water_hresp": dict(waterHResp=0),
    "anaerobic": dict(anaerobic=1),
    "anaerobic_litter": dict(anaerobic=1, litterPool=1),
    "flooding": dict(flooding=1),
    "carbon_saturation": dict(litterPool=1, carbonSaturation=1),
    "soil_phenol": dict(gdd=0, soilPhenol=1),
    "calendar_phenology": dict(gdd=0),
    "no_events": dict(events=0),
    "nitrogen": dict(litterPool=1, anaerobic=1, nitrogenCycle=1),
    "everything": dict(litterPool=1, anaerobic=1, nitrogenCycle=1, carbonSaturation=1, flooding=1, growthResp=1, leafWater=1),
    "russell_3": dict(growthResp=1, leafWater=1, litterPool=1, waterHResp=0),
}
SHAPES = {"c10k": (1, 10240), "c4": (32, 1024), "c3f32": (1, 65536), "c10krec": (1, 10240)}
T = 17520
out = sys.argv[1] if len(sys.argv) > 1 else None
shapes = sys.argv[2:] or ["c10k", "c4"]
rows = []
for shape in shapes:
    S, M = SHAPES[shape]
    clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(S)]
    base_ms = None
    for name, kw in SETS.items():
        flags = sa.flags_from(**kw)
        base, _ = sa.read_params(os.path.join(REPO, "tests", "golden", "synth", "allflags.param"), flags)
        members = synth.perturbed_params(base, M)
        f32, rec = shape == "c3f32", shape == "c10krec"
        b = sa.Batch(flags, S, M, sa.F32_MIXED if f32 else sa.F64, fast_math=None if f32 else True)
        for s in range(S):
            b.set_climate(s, clims[s])
        b.set_params(sa._lib.ALL_SITES, members)
        planes, recbuf = b.alloc_outputs(T, full=rec)
        ms = []
        for _ in range(3):
            b.setup()
            if rec:
                b.run(0, T, planes=planes, rec=recbuf)
            else:
                b.run(0, T, planes=planes)
            torch.cuda.synchronize()
            ms.append(b.last_kernel_ms())
        k = b.last_launch()["kernel"]
        b.close()
        m = min(ms)
        if name == "default":
            base_ms = m
        rows.append((shape, name, k, m, m / base_ms))
        print(f"{shape:5s} {name:20s} {k:48s} {m:8.2f} ms  x{m / base_ms:5.2f}  {S * M * T / m / 1e6:7.2f} G steps/s", flush=True)
if out:
    with open(out, "w") as f:
        f.write("| shape | flag set | kernel (AUTO) | ms / launch | vs default | G steps/s |\n|---|---|---|---|---|---|\n")
        for shape, name, k, m, r in rows:
            S, M = SHAPES[shape]
            f.write(f"| {shape} | {name} | `{k}` | {m:.2f} | {r:.2f} | {S * M * T / m / 1e6:.1f} |\n")
