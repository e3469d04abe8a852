#!/usr/bin/env python3
"""dev: the device-built site plan's two sequential walks, per forcing shape (wall_clock64 stamps of planSeqKernel).
usage: plan_device_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from tests import helpers
from tests.test_gpu_plan_device import FORCINGS, compare
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), sa.flags_from())
for name, clim in FORCINGS.items():
    b = sa.Batch(sa.flags_from(), 1, 64, sa.F64, fast_math=True, kernel_options=sa.KOPT_DEVICE_PLAN)
    b.set_climate(0, clim); b.set_params(None, base); b.setup()
    r = compare(b, 0)
    print(f"{name:36s} {clim.n_steps:6d} records  runs {r['runs']:3d}  evictions {r['n_ops']:6d}  ring walk {r['ring_walk_us']:8.1f} us  differing records {r['records']}")
    b.close()
