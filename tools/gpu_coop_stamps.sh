#!/bin/bash
# dev: where does the carbon wave of the cooperative kernel spend a step?  (-DSIPNET_STAMPS build)
cd "$GRAFT_REPO_ROOT/sipnet_amd/csrc" || exit 1
cp ../libsipnet_amd.so /tmp/lib_orig.so
for e in 0; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc -DSIPNET_STAMPS -c step_coop.hip -o /tmp/step_coop_s.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipnet_amd.so engine.o step_kernel.o step_fast.o /tmp/step_coop_s.o pf.o plan.o host_io.o restart_io.o || exit 1
(cd ../..; SIPNET_COOP=1 python3 - <<PY
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch, sipnet_amd as sa
from sipnet_amd import synth
os.environ["SIPNET_FAST_MATH"] = "1"
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
T, M = 17520, 10240
b = sa.Batch(flags, 1, M, sa.F64)
b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
b.set_params(0, synth.perturbed_params(base, M))
b.setup(); planes, _ = b.run(); torch.cuda.synchronize()
st = (C.c_ulonglong * 8)()
sa.lib().sipnet_debug_read_coop_stamps.argtypes = [C.c_void_p]
sa.lib().sipnet_debug_read_coop_stamps(st)
v = np.array(list(st), dtype=float)
names = ["loop+record+factors take", "fluxes", "events test", "take psn", "pools+mortality+post lai", "soilC+outputs", "ring + stores"]
print("kernel ms", b.last_kernel_ms(), "C-wave cycles/step (100 MHz ticks x24)", v.sum() / T)
for n, x in zip(names, v): print("  %-34s %8.1f cycles/step" % (n, x / T))
PY
)
done
cp /tmp/lib_orig.so ../libsipnet_amd.so
