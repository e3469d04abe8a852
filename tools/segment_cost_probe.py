"""What a launch boundary costs: one year of c10k's shape (or members/sites given) in k launches on one stream, against
one launch; the kernel's own time per launch from the library's HIP events.
usage: python tools/segment_cost_probe.py [members=10240] [sites=1] [prec=f64|f32]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sipnet_amd as sa  # noqa: E402
from sipnet_amd import synth  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sa.F32_MIXED if len(sys.argv) > 3 and sys.argv[3] == "f32" else sa.F64
T = 17520
flags = sa.flags_from()
base = sa.read_params("sipnet_amd/data/base_forest.param", flags)[0]
b = sa.Batch(flags, S, M, prec, fast_math=True)
for s in range(S):
    b.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
b.set_params(None, synth.perturbed_params(base, M))
b.setup()
planes, _ = b.alloc_outputs(T)
b.run(0, T, planes=planes)
torch.cuda.synchronize()
print("kernel", b.last_launch()["kernel"], "grid", b.last_launch()["grid"])
for k in (1, 2, 4, 10, 20, 40, 365):
    cuts = [((T * j // k) // 16) * 16 for j in range(k)] + [T]
    ts, ks = [], []
    for rep in range(4):
        b.setup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, z in zip(cuts[:-1], cuts[1:]):
            b.time_next_launch(); b.run(a, z - a, planes=planes[:, a:z])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(((time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3))
    ts.sort()
    last_ms = b.last_kernel_ms()
    print("%3d launch(es): %8.3f ms wall (host enqueue %.3f ms); last launch's kernel %.4f ms for %d steps = %.2f us/step"
          % (k, ts[1][0], ts[1][1], last_ms, cuts[-1] - cuts[-2], last_ms * 1e3 / (cuts[-1] - cuts[-2])))
b.close()
