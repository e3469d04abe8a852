"""probe: device-side time of (setup + run) passes right after a device synchronisation (is the first one slower?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), 'data', 'base_forest.param'), flags)
M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 17520
b = sa.Batch(flags, 1, M, sa.F64, fast_math=True)
b.set_climate(0, synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T))))
b.set_params(0, synth.perturbed_params(base, M))
b.setup()
planes, _ = b.alloc_outputs(T)
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    for i in range(7):
        if i == 2:
            torch.cuda.synchronize()
            t_sync = time.perf_counter()
        ev[i].record()
        b.setup(); b.run(0, T, planes=planes)
    ev[7].record()
    torch.cuda.synchronize()
    print(rep, "device ms per pass:", " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(7)), " wall of the 5 passes after the sync: %.3f ms per pass" % ((time.perf_counter() - t_sync) * 1e3 / 5), flush=True)
