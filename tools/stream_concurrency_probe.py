"""Do two step kernels on two streams of one process share the device?  K batches of M members each (one site, one
year), launched back to back on K streams; the time of all of them against the time of one.
usage: python tools/stream_concurrency_probe.py [members=1024] [batches=2]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import sipnet_amd as sa  # noqa: E402
from sipnet_amd import synth  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = 17520
flags = sa.flags_from()
base = sa.read_params("sipnet_amd/data/base_forest.param", flags)[0]
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
members = synth.perturbed_params(base, M)
lanes = []
for k in range(K):
    b = sa.Batch(flags, 1, M, sa.F64, fast_math=True)
    b.set_climate(0, clim)
    b.set_params(0, members)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.setup()
        planes, _ = b.alloc_outputs(T)
        b.run(0, T, planes=planes)
    lanes.append((b, s, planes))
torch.cuda.synchronize()


def go(n):
    ts = []
    for _ in range(4):
        for b, s, _ in lanes[:n]:
            with torch.cuda.stream(s):
                b.setup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b, s, p in lanes[:n]:
            with torch.cuda.stream(s):
                b.run(0, T, planes=p)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[1]


li = lanes[0][0].last_launch()
print("kernel %s, grid %d, lds %d B" % (li["kernel"], li["grid"], li["lds_bytes"]))
for n in range(1, K + 1):
    print("%d batch(es) of %d members on %d stream(s): %.3f ms" % (n, M, n, go(n)))
