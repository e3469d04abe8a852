#!/usr/bin/env python3
"""dev (GPU box): the drop-in CLI on an ensemble over a half-hourly year -- the members' text files against the
ensemble output block (`--ensemble-out`: three planes from the lean throughput kernels, or named `.out` columns from
the record).  usage: cli_block_time.py [members_block=10240] [members_text=512]"""
import os, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from sipnet_amd import synth

CLI = os.path.join(REPO, "sipnet_amd", "bin", "sipnet")
MB, MT = (int(sys.argv[1]) if len(sys.argv) > 1 else 10240), (int(sys.argv[2]) if len(sys.argv) > 2 else 512)
T = 17520
tmp = tempfile.mkdtemp(prefix="cli_blk_", dir="/tmp")
synth.write_clim(os.path.join(tmp, "sipnet.clim"), synth.round_like_file(synth.half_hourly_year_raw(T)))
shutil.copyfile(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), os.path.join(tmp, "sipnet.param"))
open(os.path.join(tmp, "sipnet.in"), "w").write("EVENTS = 0\nDO_MAIN_OUTPUT = 1\nPRINT_HEADER = 1\n")
rng = np.random.default_rng(1)


def table(M):
    with open(os.path.join(tmp, "members.txt"), "w") as f:
        f.write("aMax psnTOpt baseVegResp\n")
        for m in range(M):
            f.write(f"{9.0 + 0.4 * rng.standard_normal():.6f} {24 + rng.standard_normal():.6f} {0.010 * np.exp(0.1 * rng.standard_normal()):.8f}\n")


def run(label, M, *args):
    for f in os.listdir(tmp):
        if f.endswith((".out", ".nc")):
            os.remove(os.path.join(tmp, f))
    t0 = time.time()
    r = subprocess.run([CLI, "-i", "sipnet.in", "--ensemble-params", "members.txt", *args], cwd=tmp, capture_output=True, text=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for line in r.stdout.splitlines():
        if "ensemble block:" in line or "44-column record in" in line:
            print("      " + line.strip())
    size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.endswith((".out", ".nc")))
    n = sum(1 for f in os.listdir(tmp) if f.endswith((".out", ".nc")))
    print(f"{label}: {M} members x {T} steps, CLI wall {dt:.2f} s, {n} output file(s), {size/1e9:.3f} GB, "
          f"{M*T/dt/1e6:.1f} M member-steps/s end to end", flush=True)
    return dt


table(MT)
t_text = run("members' text files", MT)
print(f"   (extrapolated to {MB} members: {t_text * MB / MT:.0f} s, {MB} files)")
table(MB)
run("block, three planes (f64)", MB, "--ensemble-out", "ens.nc")
run("block, three planes (f32)", MB, "--ensemble-out", "ens.nc", "--ensemble-out-f32")
run("block, DAILY SUMS of the three planes (f64; --ensemble-out-sums 48)", MB, "--ensemble-out", "ens.nc", "--ensemble-out-sums", "48")
run("block, 8 columns of the record (f32)", MB, "--ensemble-out", "ens.nc", "--ensemble-out-f32", "--ensemble-out-columns",
    "nee,gpp,evapotranspiration,plantWoodC,plantLeafC,soil,soilWater,snow")
shutil.rmtree(tmp)
