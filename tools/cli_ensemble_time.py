#!/usr/bin/env python3
"""dev: wall time of the drop-in CLI on an ensemble (niwot forcing, M members, one batch)"""
import os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from tests import helpers
from tests.test_cli import CLI, stage
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tmp = tempfile.mkdtemp(prefix="cli_ens_")
stage("niwot", tmp)
rng = np.random.default_rng(1)
with open(os.path.join(tmp, "members.txt"), "w") as f:
    f.write("aMax psnTOpt\n")
    for m in range(M):
        f.write(f"{8.3 + 0.3 * rng.standard_normal():.6f} {24 + rng.standard_normal():.6f}\n")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([CLI, "-i", "sipnet.in", "--ensemble-params", "members.txt"], cwd=tmp,
                       capture_output=True, text=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp) if f.endswith(".out"))
    print(f"{M} members x 5237 steps: CLI wall {dt:.2f} s, {size/1e6:.0f} MB of .out text ({size/1e6/dt:.0f} MB/s), "
          f"{M*5237/dt/1e6:.2f} M member-steps/s end to end", flush=True)
subprocess.run(["rm", "-rf", tmp])
