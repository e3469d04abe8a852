#!/bin/bash
# dev: time c10k with step_coop.hip rebuilt with extra -D flags (experiments; results may be wrong)
# usage: gpu_coop_exp.sh "-DEXP_A" "-DEXP_B" ...
cd "$GRAFT_REPO_ROOT/sipnet_amd/csrc" || exit 1
cp ../libsipnet_amd.so /tmp/lib_orig.so
for e in "" "$@"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc $e -c step_coop.hip -o /tmp/step_coop_x.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipnet_amd.so engine.o step_kernel.o step_fast.o /tmp/step_coop_x.o pf.o plan.o host_io.o restart_io.o || exit 1
(cd ../..; python3 - "$e" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["SIPNET_FAST_MATH"] = "1"
import torch, sipnet_amd as sa
from sipnet_amd import synth
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
T, M = 17520, 10240
b = sa.Batch(flags, 1, M, sa.F64)
b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
b.set_params(0, synth.perturbed_params(base, M))
ms = []
for r in range(3):
    b.setup(); planes, _ = b.run(); torch.cuda.synchronize(); ms.append(b.last_kernel_ms())
print("variant [%s]: %.2f ms (min of 3), NEE checksum %.10g" % (sys.argv[1], min(ms), float(planes[0].double().sum())), flush=True)
PY
)
done
cp /tmp/lib_orig.so ../libsipnet_amd.so
