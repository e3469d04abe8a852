#!/usr/bin/env python3
"""dev: ten simulated years (175 200 half-hourly steps, the synthetic year repeated with new
weather noise each year) of 10 240 members on the throughput kernel, run year by year into one
plane buffer; eight members followed by the oracle over the whole decade."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
from tests import helpers
YEARS, M = 10, 10240
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), "data", "base_forest.param"), flags)
raw = synth.half_hourly_year_raw(17520 * YEARS)
clim = synth.convert_raw(synth.round_like_file(raw))
members = synth.perturbed_params(base, M)
b = sa.Batch(flags, 1, M, sa.F64, fast_math=True)
b.set_climate(0, clim); b.set_params(0, members); b.setup()
planes, _ = b.alloc_outputs(17520)
pick = [0, 1, 2, 3, 5000, 5001, 10238, 10239]
got = np.zeros((3, 17520 * YEARS, len(pick)))
torch.cuda.synchronize(); t0 = time.perf_counter(); kms = 0.0
for y in range(YEARS):
    b.run(y * 17520, 17520, planes=planes)
    torch.cuda.synchronize(); kms += b.last_kernel_ms()
    got[:, y * 17520:(y + 1) * 17520] = planes[:, :, pick].cpu().numpy()
wall = time.perf_counter() - t0
state = b.get_state(); b.close()
oracle = helpers.load_oracle()
want, final, st = oracle.run_block(flags, members[pick], clim)
err = np.abs(got - want)
print(f"{YEARS} years x {M} members: kernel {kms:.1f} ms ({M*17520*YEARS/kms/1e6:.2f} G steps/s), wall incl. sampling copies {wall*1e3:.0f} ms")
for y in (0, 4, 9):
    sl = slice(y * 17520, (y + 1) * 17520)
    print(f"  year {y+1:2d}: max|dNEE| {err[0, sl].max():.2e}  max|dGPP| {err[1, sl].max():.2e}  max|dET| {err[2, sl].max():.2e}")
pools = np.abs(state[pick, :13] - final[:, 14:27]) / np.maximum(np.abs(final[:, 14:27]), 1e-2)
print(f"  pools after {YEARS} years: max relative difference {pools.max():.2e}; all finite: {bool(np.isfinite(state).all())}")
