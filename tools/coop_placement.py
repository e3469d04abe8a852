#!/usr/bin/env python3
"""dev (GPU box): where the cooperative kernel's wavefronts run.  Needs the diagnostic build
`python tools/build_variants.py hwid=-DSIPNET_HWID`.  For a workload prints, per role, how the
waves spread over the four SIMDs of their CU, how often the three (four) waves of a chunk sit on
as many different SIMDs, and -- for co-resident workgroups -- how often two CARBON waves (the
critical ones) share a SIMD.   usage: coop_placement.py [workload] [kernel]"""
import collections, ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sipnet_amd import _lib
_lib.use_library(os.path.join(REPO, "build", "variants", "hwid", "libsipnet_amd.so"))
import numpy as np, torch, sipnet_amd as sa
from sipnet_amd import synth
from bench import WORKLOADS
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c10k"]
kern = dict(auto=sa.KERNEL_AUTO, coop_hbm=sa.KERNEL_COOP_HBM, coop_lds=sa.KERNEL_COOP_LDS, coop_pair=sa.KERNEL_COOP_PAIR, coop_quad=sa.KERNEL_COOP_QUAD)[sys.argv[2] if len(sys.argv) > 2 else "auto"]
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
S, M, T = wl["sites"], wl["members"], 48 * 20
prec = sa.F64 if wl["prec"] == "f64" else sa.F32_MIXED
b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None, kernel=kern)
for s in range(S):
    b.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
    b.set_params(s, synth.perturbed_params(base, M))
b.setup(); b.run(); torch.cuda.synchronize()
li = b.last_launch()
out = (C.c_uint * (4096 * 4 * 2))()
_lib.lib().sipnet_debug_read_coop_hwid(out)
per_wg = 2 if "Pair" in li["kernel"] else 4 if "Quad" in li["kernel"] else 1        # chunks per workgroup
nroles = 4 if li["block_threads"] == 256 else 3                                     # with the factor wave or without
a = np.frombuffer(out, dtype=np.uint32).reshape(4096, 4, 2)[:min(li["grid"] * per_wg, 4096), :nroles]
hw, xcc = a[..., 0], a[..., 1] & 0xf
simd = (hw >> 4) & 3
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc.astype(np.uint32) << 8)
print(li["kernel"], "grid", li["grid"])
for r, name in enumerate(("carbon", "water", "light", "factors")[:nroles]):
    print(f"  {name:6s} waves by SIMD:", np.bincount(simd[:, r], minlength=4).tolist())
distinct = sum(len(set(simd[g])) == nroles for g in range(len(a)))
print(f"  chunks with their {nroles} waves on {nroles} different SIMDs: {distinct} of {len(a)}")
bycu = collections.defaultdict(list)
for g in range(len(a)):
    bycu[int(cu[g, 0])].append(int(simd[g, 0]))
share = sum(len(v) - len(set(v)) for v in bycu.values())
print(f"  CUs used: {len(bycu)}; chunks per CU: {collections.Counter(len(v) for v in bycu.values())}; "
      f"carbon waves that share a SIMD with another carbon wave of their CU: {share}")
