#!/usr/bin/env python3
"""probe (GPU): what the forecast's parameter reads cost once a filter has resampled for a while -- the one-wave kernel's 48-step
launch at c5's shape (131 072 fp32-mixed particles) timed by the library's HIP events: fresh ensemble (parameters in column
order), after 30 cycles through the index into the batch's own block (a one-rank filter), through the index into the replicated
bank of a pretended world of 8, and with SIPNET_KOPT_PF_MOVE_PARAMS (rows moved: column order again).
usage: forecast_param_reads.py [cycles=30]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth, dist as sd

n, T = 131072, 48
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 30
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(REPO, "sipnet_amd", "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T * 40)))
members = synth.perturbed_params(base, n)


def forecast_ms(b, planes, k):
    ms = []
    for r in range(5):
        b.time_next_launch()
        b.run((k % 40) * T, T, planes=planes)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    return min(ms) * 1e3


for tag, opt, world in (("one rank, index into its own block", 0, 1), ("pretended world of 8, index into the replicated bank", 0, 8),
                        ("pretended world of 8, parameter rows move", sa.KOPT_PF_MOVE_PARAMS, 8)):
    b = sa.Batch(flags, 1, n, sa.F32_MIXED, kernel=sa.KERNEL_ONE_WAVE, kernel_options=opt)
    b.set_climate(0, clim)
    b.set_params(0, members)
    b.setup()
    planes, _ = b.alloc_outputs(T)
    fresh = forecast_ms(b, planes, 0)
    b.setup()
    if world > 1:
        sd.pf_connect_peers(b, 0, 1, with_params=True, pretend_world=world)
    for k in range(cycles):
        b.run(k * T, T, planes=planes)
        tot = planes[0].double().sum(0)
        obs, sigma = float(tot.median()), float(tot.std()) * 1.5 + 1e-12
        if world > 1:
            sd.pf_analysis_peers(b, planes[0], obs, sigma, 0.5, collectives=False, pretend_world=world)
        else:
            sd.pf_analysis(b, planes[0], obs, sigma, u0=0.5, with_params=True, diagnostics=False, collectives=False)
    after = forecast_ms(b, planes, cycles)
    print("%-58s forecast kernel: fresh %.1f us, after %d cycles %.1f us" % (tag, fresh, cycles, after), flush=True)
    b.close()
