#!/usr/bin/env python3
"""dev: c10k end to end INCLUDING the host<->HBM transfers at the boundary (DESIGN.md section 4
quotes these; bench.py's `value` is the HBM-resident rate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sipnet_amd as sa
from sipnet_amd import synth
T, M = 17520, 10240
flags = sa.flags_from()
base, _ = sa.read_params(os.path.join(os.path.dirname(sa.__file__), "data", "base_forest.param"), flags)
clim = synth.convert_raw(synth.round_like_file(synth.half_hourly_year_raw(T)))
members = synth.perturbed_params(base, M)
torch.zeros(1, device="cuda")
sync = torch.cuda.synchronize
for rep in range(3):
    t0 = time.perf_counter()
    b = sa.Batch(flags, 1, M, sa.F64, fast_math=True)  # what bench.py times
    b.set_climate(0, clim); b.set_params(0, members); b.setup(); sync()
    t_in = time.perf_counter() - t0
    planes, _ = b.alloc_outputs(T); sync()
    t0 = time.perf_counter(); b.run(0, T, planes=planes); sync(); t_run = time.perf_counter() - t0
    stats = torch.empty((3, T, 1, 2), dtype=torch.float64, device="cuda")
    t0 = time.perf_counter()
    for v in range(3): b.reduce_plane(planes[v], stats[v])
    h = stats.cpu(); t_stats = time.perf_counter() - t0
    pinned = torch.empty(planes.shape, dtype=planes.dtype, pin_memory=True)
    t0 = time.perf_counter(); pinned.copy_(planes, non_blocking=True); sync(); t_pin = time.perf_counter() - t0
    t0 = time.perf_counter(); hp = planes.cpu(); t_page = time.perf_counter() - t0
    b.close()
    units = M * T
    gb = planes.numel() * 8 / 1e9
    print(f"rep {rep}: inputs host->HBM + plan build + setupModel {t_in*1e3:.1f} ms | step kernel {t_run*1e3:.1f} ms | "
          f"statistics (3 reductions + 0.84 MB D2H) {t_stats*1e3:.2f} ms | planes D2H {gb:.2f} GB: pinned {t_pin*1e3:.0f} ms "
          f"({gb/t_pin:.1f} GB/s), pageable {t_page*1e3:.0f} ms", flush=True)
    print(f"        G steps/s: HBM-resident {units/t_run/1e9:.2f} | + inputs {units/(t_in+t_run)/1e9:.2f} | + inputs + statistics to host "
          f"{units/(t_in+t_run+t_stats)/1e9:.2f} | + inputs + full planes to pinned host {units/(t_in+t_run+t_pin)/1e9:.2f}", flush=True)
