import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(17520)))
members = synth.perturbed_params(base, 192)
T = clim.n_steps
outs = {}
for opt in (0, sa.KOPT_NO_REGULAR_TILES):
    b = sa.Batch(flags, 1, 192, sa.F64, fast_math=True, kernel=sa.KERNEL_COOP_LDS, kernel_options=opt)
    b.set_climate(0, clim); b.set_params(0, members); b.setup()
    planes, _ = b.run()
    outs[opt] = planes.cpu().numpy(); st = b.get_state(); outs[(opt, 's')] = st
    b.close()
d = outs[0] != outs[8]
print("differing elements per plane:", d.reshape(3, -1).sum(1), "of", d[0].size)
if d.any():
    pl, t, m = np.argwhere(d)[0]
    print("first difference: plane", pl, "step", t, "member", m, outs[0][pl, t, m], outs[8][pl, t, m])
    first_t = np.argwhere(d.any(axis=(0, 2)))[0][0]
    print("first step with any difference:", first_t)
ds = outs[(0, 's')] != outs[(8, 's')]
print("state rows differing:", np.nonzero(ds.any(0))[0])
