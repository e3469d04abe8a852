import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
sys.path.insert(0, "tests")
from tests.test_gpu_batch import make_batch, BASE
base = sa.read_params(BASE, sa.flags_from())[0]
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(48 * 60)))
ev = []
def add(day, typ, *p):
    e = sa.Event(); e.type = typ; e.year = int(clim.year[0]); e.day = day
    for i, v in enumerate(p): e.p[i] = v
    ev.append(e)
add(4, 2, 1.5, 0); add(9, 1, 0.3, 0.2, 0.1, 0.1); add(20, 1, 1.0, 1.0, 0.0, 0.0); add(30, 3, 40.0, 300.0, 50.0, 60.0); add(41, 2, 2.0, 1)
members = synth.perturbed_params(base, 150)
members[5, pi("plantWoodInit")] = 0.0
out = {}
for coop, kern in (("1", sa.KERNEL_COOP_LDS), ("2", sa.KERNEL_COOP_HBM), ("3", sa.KERNEL_COOP_PAIR), ("4", sa.KERNEL_COOP_QUAD)):
    b = make_batch(sa.flags_from(), [clim], members, prec=sa.F32_MIXED, events=ev, kernel=kern)
    T = clim.n_steps
    planes, _ = b.alloc_outputs(T)
    for a, z in ((0, 7), (7, 1000), (1000, 1015), (1015, T)):
        b.run(a, z - a, planes=planes[:, a:z])
    out[coop] = planes.cpu().numpy().astype(np.float64)
    b.close()
for m in "234":
    d = out["1"] != out[m]
    print("mode", m, "mismatches per plane", d.sum(axis=(1, 2)))
    if d.any():
        p, t, c = np.nonzero(d)
        print("  first steps", t[:10], "members", c[:10], "planes", p[:10])
        print("  steps range", t.min(), t.max(), "unique steps", len(np.unique(t)), "unique members", len(np.unique(c)))
        print("  values", out["1"][p[0], t[0], c[0]], out[m][p[0], t[0], c[0]])
