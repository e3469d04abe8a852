"""What the overlapped plane gather of the C host costs (sipnet_node_run_gathering, csrc/node.cpp): one year on a
node of SHARDS shards of one device each listed in DEVICES, plain run (planes + statistics, nothing gathered)
against run_gathering in 1 / 4 / 10 / 20 segments, ms per pass after a warm-up pass.  On a one-GPU box the shards
share device 0, so the gathers are device-to-device copies through the same HBM the kernels use -- an upper
bound of the interference, not a link measurement.
usage: python tools/node_gather_time.py [members=10240] [sites=1] [devices=0,0] [members|sites]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import sipnet_amd as sa  # noqa: E402
from sipnet_amd import synth  # noqa: E402
from sipnet_amd._lib import SHARD_MEMBERS, SHARD_SITES  # noqa: E402
from sipnet_amd.node import Node  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
devices = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,0").split(",")]
shard = SHARD_SITES if len(sys.argv) > 4 and sys.argv[4] == "sites" else SHARD_MEMBERS
T = 17520
flags = sa.flags_from()
base = sa.read_params("sipnet_amd/data/base_forest.param", flags)[0]
members = synth.perturbed_params(base, M)
nd = Node(flags, S, M, devices=devices, shard=shard, fast_math=True)
for s in range(S):
    nd.set_climate(s, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T, site=s))))
nd.set_params(None, members)


def timed(f, reps=3):
    ts = []
    for r in range(reps + 1):
        nd.setup()
        nd.sync()
        t0 = time.perf_counter()
        f()
        nd.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[1:]))


plain = timed(lambda: nd.run(0, T))
print("node: %d shard(s) on devices %s, %s sharded, %d site(s) x %d members x %d steps, kernel %s, transport %s" % (
    nd.n, devices, "sites" if shard == SHARD_SITES else "members", S, M, T, nd.kernel_name(0), nd.collective_library()))
fc = timed(lambda: nd.forecast(0, T))
print("run (planes + statistics, nothing gathered)   %8.3f ms" % plain)
print("forecast (planes only)                        %8.3f ms" % fc)
gb = nd.n * 3 * T * nd.ld * (8 if nd.precision == sa.F64 else 4) / 1e9


def one_gather():
    nd.forecast(0, T)
    nd.gather_planes()


print("forecast + gather_planes afterwards           %8.3f ms   (%.2f GB received per device)" % (timed(one_gather), gb))
for nseg in (1, 4, 10, 20, 40):
    ms = timed(lambda: nd.run_gathering(0, T, nseg))
    print("run_gathering, %2d segments                    %8.3f ms   (+%.3f over the forecast)" % (nseg, ms, ms - fc))
# the member-resolved exchange in the forms that fit under the kernel (sipnet_node_run_gathering_reduced)
for form, k, what in (("sums", 48, "daily sums per member (doubles)"), ("f32", 0, "the planes as floats")):
    if form == "f32" and nd.precision != sa.F64:
        continue
    rows = (T + k - 1) // k if k else T
    mb = 3 * rows * nd.ld * (8 if form == "sums" else 4) / 1e6
    for nseg in (1, 4, 8, 16):
        ms = timed(lambda: nd.run_gathering_reduced(0, T, nseg, form, k))
        print("run_gathering_reduced %-4s, %2d segments         %8.3f ms   (+%.3f over the forecast; %s: %.1f MB per rank)" % (
            form, nseg, ms, ms - fc, what, mb))
nd.close()
