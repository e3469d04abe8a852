#!/usr/bin/env python3
"""Distil a gpurun_out/prof_<tag>_<workload>/ directory (written by tools/gpu_profile.sh)
into committed evidence under profiles/: the rocprofv3 --kernel-trace --stats table, the
PMC means per kernel, and the HBM traffic per launch with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports 1/2 of the
bytes of a coalesced streaming read -- checked here against reducePlaneKernel, whose
read volume is known exactly)."""
import collections, csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1].rstrip("/")
name = os.path.basename(src).replace("prof_", "")
tag, wl = name.rsplit("_", 1)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

def one(pattern):
    # (gpurun MERGES a call's output into gpurun_out/: an earlier call's files, with other process ids in
    # their names, may still be there -- take the newest)
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None

out = [f"# rocprofv3 summary -- {tag}, workload {wl}", ""]
ks = one("trace/**/*kernel_stats.csv")
if ks:
    shutil.copyfile(ks, os.path.join(dst, f"{name}_kernel_stats.csv"))
    out += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --steps 5 --warmup 1 --no-cpu-baseline --no-fill-probe --no-end-to-end`" % wl, "",
            "| kernel | calls | total ns | avg ns | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks)):
        out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |")
    out.append("")
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith("{"):
            j = json.loads(line)
            out += ["bench.py line under the profiler: value %.4g %s, ms_per_step %.3f, roofline.kernel_ms %.3f (HIP events)" % (
                j["value"], j["unit"], j["ms_per_step"], j["roofline"]["kernel_ms"]), ""]

means = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = one(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        means[(k, c)] = sum(v) / len(v)
if means:
    out += ["## PMC (separate `--pmc` passes, tools/prof_target.py: 2 x [setup, step kernel, reducePlane])", "",
            "| kernel | counter | mean per dispatch |", "|---|---|---|"]
    for (k, c), v in sorted(means.items()):
        if "sipnet" in k:
            out.append(f"| `{k[:70]}` | {c} | {v:.6g} |")
    out.append("")
    # the step kernel: the instantiation the library says it launched (tools/prof_target.py prints
    # sipnet_batch_last_launch's name), else any step*Kernel of this engine
    import re
    launched = None
    try:
        for line in open(os.path.join(src, "pmc_fetch.log")):
            if line.startswith("step kernel name"):
                launched = line.split("name", 1)[1].strip()
    except OSError:
        pass
    norm = lambda k: k.replace("void sipnet::", "").replace("(anonymous namespace)::", "").split("(sipnet::")[0].strip()
    step = [k for (k, c) in means if launched and norm(k) == launched]
    if not step:
        step = [k for (k, c) in means if re.search(r"\bstep(Coop[A-Za-z]*|Fast)?Kernel<", k)]
    red = [k for (k, c) in means if "reducePlane" in k]
    if step and red:
        sk, rk = step[0], red[0]
        f_step, w_step = means.get((sk, "FETCH_SIZE")), means.get((sk, "WRITE_SIZE"))
        f_red = means.get((rk, "FETCH_SIZE"))
        log = open(os.path.join(src, "pmc_fetch.log")).read()
        plane_bytes = None
        for tok in log.split("plane bytes")[1:]:
            plane_bytes = float(tok.split()[0])
        if f_step and w_step and f_red and plane_bytes:
            corr = plane_bytes / (f_red * 1024.0)
            hbm = f_step * 1024.0 * corr + w_step * 1024.0
            out += ["## HBM traffic of the step kernel, per launch", "",
                    f"* calibration: reducePlaneKernel reads exactly {plane_bytes:.0f} B; FETCH_SIZE says "
                    f"{f_red*1024:.0f} B -> read-side correction x{corr:.3f} (the guide's gfx950 1/2 factor)",
                    f"* FETCH_SIZE {f_step:.6g} KiB x {corr:.3f} = {f_step*1024*corr/1e9:.3f} GB read",
                    f"* WRITE_SIZE {w_step:.6g} KiB = {w_step*1024/1e9:.3f} GB written",
                    f"* **traffic = {hbm/1e9:.3f} GB per launch**", ""]
            tj = os.path.join(dst, "pmc_traffic.json")
            d = json.load(open(tj)) if os.path.exists(tj) else {}
            kname = norm(sk)
            d[wl] = {"hbm_bytes_per_launch": hbm, "tag": tag, "fetch_correction": corr, "kernel": kname}
            for line in log.splitlines():   # (tools/prof_target.py: the device sources the profiled library was built from)
                if line.startswith("kernel sources sha16"):
                    d[wl]["source_sha16"] = line.split()[-1]
            # the ALU cross-check of SURVEY 8(d): cycles in which a SIMD's vector ALU was executing, summed over
            # the chip (SQ_ACTIVE_INST_VALU counts quad-cycles) -- bench.py divides by SIMDs x kernel cycles
            av = means.get((sk, "SQ_ACTIVE_INST_VALU"))
            if av:
                d[wl]["valu_active_cycles_per_launch"] = av * 4.0
                if ks:
                    for r in csv.DictReader(open(ks)):
                        if norm(r["Name"]) == kname:
                            d[wl]["profiled_kernel_avg_ns"] = float(r["AverageNs"])
            json.dump(d, open(tj, "w"), indent=1)
        sq = {c: means[(sk, c)] for (k, c) in means if k == sk and c.startswith("SQ_")}
        if sq:
            out += ["## Issue statistics of the step kernel (SQ counters; *_CYCLES are quad-cycles)", ""]
            for c, v in sorted(sq.items()):
                out.append(f"* {c} = {v:.6g}")
            out.append("")
open(os.path.join(dst, f"{name}.md"), "w").write("\n".join(out))
print("\n".join(out))
