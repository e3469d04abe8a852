#!/bin/bash
# dev: cycles each of the three waves of the cooperative kernel spends in its hand-over waits
# (-DSIPNET_WAITS build; s_memtime counts at 100 MHz)
cd "$GRAFT_REPO_ROOT/sipnet_amd/csrc" || exit 1
cp ../libsipnet_amd.so /tmp/lib_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc -DSIPNET_WAITS $1 -c step_coop.hip -o /tmp/step_coop_w.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipnet_amd.so engine.o step_kernel.o step_fast.o /tmp/step_coop_w.o pf.o plan.o host_io.o restart_io.o || exit 1
(cd ../..; python3 - <<'PY'
import os, sys, ctypes as C, numpy as np
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
b.setup(); planes, _ = b.run(); torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
sa.lib().sipnet_debug_read_coop_waits.argtypes = [C.c_void_p]
sa.lib().sipnet_debug_read_coop_waits(st)
v = np.array(list(st), dtype=float)
print("kernel ms", b.last_kernel_ms())
tick = b.last_kernel_ms() * 1e-3 * 2.4e9 / max(v[3], v[7], v[11])   # cycles per s_memtime tick
for name, base_, labels in (("L", 0, ["take lai", "-", "-"]), ("W", 4, ["take pgp+alive", "progress wait", "-"]),
                            ("C", 8, ["record+factors take", "take psn", "-"])):
    tot = v[base_ + 3]
    print(f"wave {name}: total {tot*tick/T:7.0f} cycles/step; " + "; ".join(
        f"{l} {v[base_+k]*tick/T:6.0f}" for k, l in enumerate(labels) if l != "-"))
PY
)
cp /tmp/lib_orig.so ../libsipnet_amd.so
