#!/usr/bin/env python3
"""dev (GPU box): time the c10k step kernel of every library under build/variants/ (plus the
in-tree product) and check 8 members against the oracle, one child process per library.

usage: variant_bench.py [--workload c10k] [--reps 5] [--kernel auto|one_wave|coop_lds|coop_hbm|coop_pair] [names...]
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
lib, wl_name, reps, kern, kopt = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
from sipnet_amd import _lib
if lib != "product":
    _lib.use_library(lib)
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from bench import WORKLOADS
from tests import helpers
wl = WORKLOADS[wl_name]
flags = sa.flags_from(**wl.get("flags", {}))
base, _ = sa.read_params(os.path.join(sys.argv[1], "sipnet_amd", "data", wl.get("param", "base_forest.param")), flags)
S, M, T = wl["sites"], wl["members"], wl["steps"]
prec = sa.F64 if os.environ.get("VB_PREC", wl["prec"]) == "f64" else sa.F32_MIXED
K = dict(auto=sa.KERNEL_AUTO, one_wave=sa.KERNEL_ONE_WAVE, coop_lds=sa.KERNEL_COOP_LDS, coop_hbm=sa.KERNEL_COOP_HBM, coop_pair=sa.KERNEL_COOP_PAIR, coop_quad=sa.KERNEL_COOP_QUAD)[kern]
b = sa.Batch(flags, S, M, prec, fast_math=True if prec == sa.F64 else None, kernel=K, kernel_options=kopt)
members = synth.perturbed_params(base, M)
clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(S)]
for s in range(S):
    b.set_climate(s, clims[s]); b.set_params(s, members)
planes, _ = b.alloc_outputs(T)
ms = []
for r in range(reps):
    b.setup(); b.run(0, T, planes=planes); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
ora = helpers.load_oracle()
want, _, _ = ora.run_block(flags, members[:8], clims[0])
got = planes[:, :, :8].double().cpu().numpy()
d = np.abs(got - want)
print(json.dumps(dict(kernel=b.last_launch()["kernel"], ms_min=min(ms), ms_med=float(np.median(ms)),
                      dNEE=float(d[0].max()), dGPP=float(d[1].max()), dET=float(d[2].max()))))
"""

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c10k")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--kopt", type=int, default=0, help="SIPNET_KOPT_* bits (8 = no regular tiles)")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    vdir = os.path.join(REPO, "build", "variants")
    names = args.names or (["product"] + sorted(os.listdir(vdir)) if os.path.isdir(vdir) else ["product"])
    for n in names:
        lib = "product" if n == "product" else os.path.join(vdir, n, "libsipnet_amd.so")
        fl = "" if n == "product" else open(os.path.join(vdir, n, "FLAGS")).read().strip()
        r = subprocess.run([sys.executable, "-c", CHILD, REPO, lib, args.workload, str(args.reps), args.kernel, str(args.kopt)],
                           capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])
        print(f"{n:24s} [{fl}] {line}", flush=True)
