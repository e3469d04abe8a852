#!/usr/bin/env python3
"""dev: from a rocprofv3 --kernel-trace CSV of a particle-filter run (bench.py --workload c5 ...), the kernels of a typical
cycle in order with their durations and the gaps in front of them (us) -- the median over the cycles that consist of the most
frequent kernel sequence among those that contain the analysis kernel.
usage: cycle_timeline.py kernel_trace.csv [anchor kernel substring, default stepFastKernel] [must-contain substring, default pfFused]"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "stepFastKernel"
must = sys.argv[3] if len(sys.argv) > 3 else "pfFused"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
cycles, cur = [], None
for s, e, n in ev:
    if anchor in n:
        if cur:
            cycles.append(cur)
        cur = []
    if cur is not None:
        cur.append((s, e, n))
if cur:
    cycles.append(cur)
def short(n):
    n = n.replace("sipnet::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]
from collections import Counter
cycles = [c for c in cycles if any(must in n for _, _, n in c)]
sig = Counter(tuple(short(n) for _, _, n in c) for c in cycles)
print("%d cycles with %r" % (len(cycles), must))
for best, cnt in sig.most_common(2):
    sel = [c for c in cycles if tuple(short(n) for _, _, n in c) == best]
    print("%d cycles with this sequence of %d kernels" % (cnt, len(best)))
    for k, name in enumerate(best):
        dur = statistics.median((c[k][1] - c[k][0]) / 1e3 for c in sel)
        gap = statistics.median((c[k][0] - c[k - 1][1]) / 1e3 for c in sel) if k else 0.0
        print("  gap %6.2f  run %7.2f  %s" % (gap, dur, name))
    per = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(sel[:-1], sel[1:]) if b[0][0] - a[0][0] < 5e6]
    if per:
        print("  anchor to anchor: median %.2f us" % statistics.median(per))
