import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from sipnet_amd.config import param_index as pi
base = sa.read_params("sipnet_amd/data/base_forest.param", sa.flags_from())[0]
M, T = 64 * 3 + 17, 48 * 5 + 11
clims = [synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))) for s in range(3)]
clims[1] = clims[1].slice(0, T - 29)
members = synth.perturbed_params(base, M, seed=3)
members[5, pi("leafAllocation")] = 0.9
members[5, pi("woodAllocation")] = 0.9
members[9, pi("plantWoodInit")] = 1e-4
members[9, pi("laiInit")] = 1e-4
def make():
    b = sa.Batch(sa.flags_from(), 3, M, sa.F64, fast_math=True, kernel=sa.KERNEL_COOP_LDS)
    b.set_climates(clims); b.set_params(None, members); b.setup(); return b
ref = make()
planes, _ = ref.run(0, T)
want = planes.cpu().numpy()
print("status", ref.get_status()[:12])
for k, cuts in ((48, [0, T]), (48, [0, 96, T]), (7, [0, 7 * 9, 7 * 20, T]), (T + 5, [0, T]), (1, [0, 33, T])):
    ws = np.zeros((3, (T + k - 1) // k, want.shape[2]))
    for t in range(T): ws[:, t // k] += want[:, t]
    b = make()
    got = np.concatenate([b.run_sums(a, z - a, k).cpu().numpy() for a, z in zip(cuts[:-1], cuts[1:])], axis=1)
    bad = np.argwhere(got[:, :, :M] != ws[:, :, :M])
    print("k", k, "cuts", cuts, "mismatches", len(bad), bad[:12].tolist())
    for v, g, c in bad[:3]:
        print("   ", v, g, c, got[v, g, c], ws[v, g, c])
    b.close()
