#!/usr/bin/env python3
"""dev: `sipnet --sites LIST --devices D` over N run directories (the reference's russell_1 staging with another aMax in
each, and every fourth one niwot: two distinct forcings, so the sites of a flag set are dealt to the devices as whole
sites): wall time, and every directory's sipnet.out against the same directory run alone on device 0.
usage: cli_sites_time.py [--sites N=16] [--devices 0] [--math fast]"""
import argparse, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.test_cli import CLI, stage

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=16)
ap.add_argument("--devices", default="0")
ap.add_argument("--math", default="fast")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="cli_sites_")
dirs = []
for k in range(a.sites):
    d = os.path.join(tmp, f"run{k:03d}")
    os.mkdir(d)
    case = "niwot" if k % 4 == 3 else "russell_1"
    stage(case, d)
    txt = [(f"aMax {7.0 + 0.05 * k}" if l.split() and l.split()[0] == "aMax" else l) for l in open(os.path.join(d, "sipnet.param")).read().splitlines()]
    open(os.path.join(d, "sipnet.param"), "w").write("\n".join(txt) + "\n")
    dirs.append(os.path.basename(d))
open(os.path.join(tmp, "runs.txt"), "w").write("\n".join(dirs) + "\n")
t0 = time.time()
r = subprocess.run([CLI, "--sites", "runs.txt", "-i", "sipnet.in", "--math", a.math, "--devices", a.devices], cwd=tmp, capture_output=True, text=True)
dt = time.time() - t0
print(f"sipnet --sites ({a.sites} run directories) --devices {a.devices} --math {a.math}: rc {r.returncode}, wall {dt:.2f} s")
print("\n".join(l for l in r.stdout.splitlines() if "run(s) in" in l or "device" in l)[:2000])
if r.returncode != 0:
    print(r.stdout[-2000:] + r.stderr[-2000:])
    sys.exit(1)
outs = {d: open(os.path.join(tmp, d, "sipnet.out"), "rb").read() for d in dirs}
bad = 0
for d in dirs[:: max(1, a.sites // 4)]:            # a sample of directories run alone
    os.remove(os.path.join(tmp, d, "sipnet.out"))
    r1 = subprocess.run([CLI, "-i", "sipnet.in", "--math", a.math], cwd=os.path.join(tmp, d), capture_output=True, text=True)
    alone = open(os.path.join(tmp, d, "sipnet.out"), "rb").read() if r1.returncode == 0 else b""
    same = alone == outs[d]
    bad += 0 if same else 1
    print(f"  {d}: alone rc {r1.returncode}, identical to the stacked run's file: {same}")
subprocess.run(["rm", "-rf", tmp])
sys.exit(1 if bad else 0)
