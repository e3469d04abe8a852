#!/usr/bin/env python3
"""dev: the device-side timeline of the LAST forcing of a (rocprofv3 --kernel-trace --memory-copy-trace of
tools/e2e_breakdown.py): copies and kernels with start / end in microseconds from the first copy of the hand-over.
usage: e2e_timeline.py [gpurun_out/r5o]"""
import csv, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5o"
k = list(csv.DictReader(open(d + "/kernel_trace.csv")))
m = list(csv.DictReader(open(d + "/memory_copy_trace.csv")))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64]) for r in k] + \
     [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in m]
ev.sort()
idx = [i for i, e in enumerate(ev) if "planPrepKernel" in e[2]][-1]
start = idx
while start > 0 and ev[start - 1][0] > ev[idx][0] - 3_000_000:
    start -= 1
t0 = ev[start][0]
run = None
for s, e, n in ev[start:idx + 12]:
    if run and run[2] == n and n.startswith("COPY") and abs((e - s) - run[4]) < 0.25 * run[4] + 3000:
        run = (run[0], e, n, run[3] + 1, run[4])
        continue
    if run:
        print(f"{(run[0]-t0)/1e3:9.1f} {(run[1]-t0)/1e3:9.1f}  {run[3]:3d} x {run[4]/1e3:8.1f} us  {run[2]}")
    run = (s, e, n, 1, e - s)
print(f"{(run[0]-t0)/1e3:9.1f} {(run[1]-t0)/1e3:9.1f}  {run[3]:3d} x {run[4]/1e3:8.1f} us  {run[2]}")
