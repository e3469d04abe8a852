#!/usr/bin/env python3
"""dev: device-memory balance over 300 create / setup / run / particle-filter / destroy cycles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth, dist as sd
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(96)))
members = synth.perturbed_params(base, 4096)
free0 = None
for i in range(300):
    b = sa.Batch(flags, 2, 4096, sa.F32_MIXED if i % 2 else sa.F64, fast_math=True)
    for s in range(2):
        b.set_climate(s, clim); b.set_params(s, members)
    b.setup(); planes, rec = b.run(full=(i % 3 == 0))
    if i % 2 == 0:
        b1 = sa.Batch(flags, 1, 4096, sa.F64, fast_math=True); b1.set_climate(0, clim); b1.set_params(0, members); b1.setup()
        p1, _ = b1.run()
        sd.pf_analysis(b1, p1[0], float(p1[0].sum(0).median()), 1.0, 0.3, with_params=True, diagnostics=False)
        b1.close()
    if i % 4 == 1:
        # round 6's allocations: the in-launch sums, a filter connected to a world of itself (the replicated parameter bank, the
        # peer tables, the barrier ring of the one-launch analysis), the node object's reduced gather
        sums = b.run_sums(0, 96, 48)
        b.setup()
        b2 = sa.Batch(flags, 1, 4096, sa.F32_MIXED, kernel=sa.KERNEL_ONE_WAVE)
        b2.set_climate(0, clim); b2.set_params(0, members); b2.setup()
        p2, _ = b2.run(0, 48)
        d = b2.pf_publish(with_params=True)
        b2.pf_connect([d] * 2, 1)
        g = torch.empty((2, b2.pf_block_len()), dtype=torch.float64, device=b2.device)
        b2.pf_local_weights(p2[0], float(p2[0].double().sum(0).median()), 1.0, g[1])
        g[0] = g[1]
        anc = torch.empty(4096, dtype=torch.int32, device=b2.device)
        tot = torch.zeros(1, dtype=torch.int64, device=b2.device)
        for _ in range(3):
            b2.pf_resample_peers(g, 0.5, anc, tot)
            b2.run(0, 48, planes=p2)
        b2.close(); del sums, p2, g, anc, tot
    if i % 8 == 3:
        from sipnet_amd.node import Node
        nd = Node(flags, 2, 2048, devices=[0, 0], fast_math=True)
        for s in range(2):
            nd.set_climate(s, clim)
        nd.set_params(None, members[:2048])
        nd.setup()
        nd.run_gathering_reduced(0, 96, 2, "sums", 48)
        nd.sync()
        nd.setup()
        nd.run_gathering(0, 96, 2)
        nd.sync()
        nd.close()
    b.close(); del planes, rec
    if i in (20, 299):
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        print(i, "free GB", free / 1e9)
        if free0 is None: free0 = free
print("leak MB over 279 iterations:", (free0 - free) / 1e6)
