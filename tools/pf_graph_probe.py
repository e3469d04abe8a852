#!/usr/bin/env python3
"""dev: can a particle-filter cycle (forecast + analysis) be captured in a HIP graph through
torch.cuda.graph, and what does a replayed cycle cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth, dist as sd
T, M = 48, 131072
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
members = synth.perturbed_params(base, M)
b = sa.Batch(flags, 1, M, sa.F32_MIXED)
b.set_climate(0, clim); b.set_params(0, members); b.setup()
planes, _ = b.alloc_outputs(T)
b.run(0, T, planes=planes)
tot = planes[0].double().sum(0)
obs, sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
total = torch.ones(1, dtype=torch.int64, device="cuda")

def cycle():
    b.run(0, T, planes=planes)
    sd.pf_analysis(b, planes[0], obs, sigma, u0=0.5, with_params=True, diagnostics=False, total_out=total)

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

b.setup(); cycle(); cycle()     # warm: scratch buffers, spare state buffers
print("eager cycle: %.3f ms" % timed(cycle, 50))
b.setup(); torch.cuda.synchronize()
state_eager = None
for _ in range(4): cycle()
torch.cuda.synchronize(); state_eager = b.get_state().copy()

b.setup(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cycle(); cycle()            # two cycles: the state double buffers are back where they started
torch.cuda.synchronize()
b.setup(); torch.cuda.synchronize()
g.replay(); g.replay(); torch.cuda.synchronize()
state_graph = b.get_state().copy()
print("state after 4 cycles, graph vs eager: max|d| =", np.abs(state_graph - state_eager).max())
print("graph replay: %.3f ms per cycle" % (timed(g.replay, 25) / 2))
