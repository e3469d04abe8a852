#!/usr/bin/env python3
"""VGPRs / SGPRs / LDS / scratch of every kernel of one translation unit, from a device-only
assembly build with the Makefile's flags (dev tool):  tools/kernel_resources.py step_coop.hip"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "step_coop.hip"
extra = sys.argv[2:]
flags = {"step_coop.hip": ["-ffp-contract=fast-honor-pragmas", "-fno-honor-nans"], "step_coop_sums.hip": ["-ffp-contract=fast-honor-pragmas", "-fno-honor-nans"],
         "step_fast.hip": ["-ffp-contract=fast", "-fno-honor-nans"]}.get(src, ["-ffp-contract=off"])
out = "/tmp/kres_" + src.replace(".", "_") + ".s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
                       "--cuda-device-only", "-S", "-w", os.path.join(REPO, "sipnet_amd", "csrc", src), "-o", out]
                      + flags + (["-DSIPNET_PROBES"] if any(e.startswith("-DSIPNET_") for e in extra) else []) + extra,
                      stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(k + r" (\d+)", body) or [None, "?"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    dn = dn.replace("sipnet::", "").replace("(anonymous namespace)::", "").split("(FastArgs")[0].replace("void ", "")
    print(f"{dn[:64]:64s} vgpr {g('.amdhsa_next_free_vgpr'):>4} accum_off {g('.amdhsa_accum_offset'):>4} sgpr {g('.amdhsa_next_free_sgpr'):>4} "
          f"lds {g('.amdhsa_group_segment_fixed_size'):>7} scratch {g('.amdhsa_private_segment_fixed_size'):>4}")
