#!/usr/bin/env python3
"""dev: build experimental variants of the product library into build/variants/<name>/ (never
over the in-tree product).  Runs in the CPU container (hipcc cross-compiles gfx950); the
variants travel to the GPU box with the snapshot and tools/variant_bench.py times them.

usage: build_variants.py NAME="-DFOO -DBAR=2" [NAME2="..."] ...
       (flags apply to step_coop.hip, step_fast.hip and fast_math.h users; "base" = no flags)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sipnet_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
        "-Wno-unused-function"]
FAST_SRCS = ["step_fast.hip", "step_coop.hip"]                      # -ffp-contract=fast
OTHER_OBJS = ["engine.o", "step_kernel.o", "step_coop_bounded.o", "step_coop_sums.o", "step_fast_sums.o", "pf.o", "plan_device.o", "plan.o", "host_io.o", "restart_io.o", "ensemble_io.o", "node.o"]


def build(name, flags):
    out = os.path.join(REPO, "build", "variants", name)
    os.makedirs(out, exist_ok=True)
    objs = []
    pf_only = flags.strip() != "" and all(f.startswith("-DSIPNET_PF_") for f in flags.split())   # (the step kernels: the product's objects)
    for src in ([] if pf_only else FAST_SRCS):
        o = os.path.join(out, src.replace(".hip", ".o"))
        contract = "-ffp-contract=fast-honor-pragmas" if src == "step_coop.hip" else "-ffp-contract=fast"
        probe = ["-DSIPNET_PROBES"] if "-DSIPNET_" in flags else []     # (the probes' master switch: csrc/coop_probes.h)
        subprocess.check_call([HIPCC] + BASE + [contract, "-fno-honor-nans"] + probe + flags.split() +
                              ["-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    others = list(OTHER_OBJS)
    if "-DSIPNET_PF_" in flags:      # particle-filter probes: pf.hip rebuilt too (the Makefile's plain flags)
        o = os.path.join(out, "pf.o")
        subprocess.check_call([HIPCC] + BASE + ["-ffp-contract=off"] + flags.split() + ["-c", os.path.join(CSRC, "pf.hip"), "-o", o])
        objs.append(o)
        others.remove("pf.o")
    if pf_only:
        others += [x.replace(".hip", ".o") for x in FAST_SRCS]
    objs += [os.path.join(CSRC, o) for o in others]
    so = os.path.join(out, "libsipnet_amd.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl"])
    open(os.path.join(out, "FLAGS"), "w").write(flags + "\n")
    return name, so


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", CSRC])          # the shared objects
    jobs = []
    for a in sys.argv[1:]:
        name, _, flags = a.partition("=")
        jobs.append((name, flags))
    with ThreadPoolExecutor(4) as ex:
        for name, so in ex.map(lambda j: build(*j), jobs):
            print("built", name, "->", os.path.relpath(so, REPO))
