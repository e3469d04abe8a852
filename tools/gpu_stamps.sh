#!/bin/bash
# diagnostic: rebuild step_fast.o with -DSIPNET_STAMPS into a scratch copy of the library and
# print where a step spends its cycles (shares only; the fenced build is slower)
cd "$GRAFT_REPO_ROOT" || exit 1
cd sipnet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc -DSIPNET_STAMPS -c step_fast.hip -o /tmp/step_fast_stamps.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipnet_amd.so engine.o step_kernel.o /tmp/step_fast_stamps.o plan.o host_io.o || exit 1
cd ../..
python3 - <<'PY'
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch, sipnet_amd as sa
from sipnet_amd import synth
os.environ["SIPNET_FAST_MATH"] = "1"
flags = sa.flags_from()
base, _ = sa.read_params("sipnet_amd/data/base_forest.param", flags)
T, M = 17520, 10240
for prec, name in ((sa.F64, "f64"),):
    b = sa.Batch(flags, 1, M, prec)
    b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
    b.set_params(0, synth.perturbed_params(base, M))
    b.setup(); planes, _ = b.run(); torch.cuda.synchronize()
    st = (C.c_ulonglong * 8)()
    sa.lib().sipnet_debug_read_stamps.argtypes = [C.c_void_p]
    print("rc", sa.lib().sipnet_debug_read_stamps(st))
    v = np.array(list(st), dtype=float)
    names = ["record fetch", "start", "potPsn+light", "water", "resp..clamps", "trackers", "ring consume (wait)", "next loads+stores"]
    print(name, "kernel ms", b.last_kernel_ms(), "cycles/step total", v.sum() / T)
    for n, x in zip(names, v):
        print("  %-22s %8.1f cycles/step  %5.1f%%" % (n, x / T, 100 * x / v.sum()))
    b.close()
PY
