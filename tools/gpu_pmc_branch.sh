#!/bin/bash
# instruction mix of a workload's step kernel: branches, instruction fetches, scalar / vector / LDS instructions,
# cycles on branches (SQ_ACTIVE_INST_MISC), per launch -> gpurun_out/pmc_branch_<wl>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
WL=${1:-c10k}
OUT=gpurun_out/pmc_branch_$WL
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 tools/prof_target.py $WL 2 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 tools/prof_target.py $WL 2 > $OUT/b.log 2>&1
python3 - $OUT <<'PY' | tee gpurun_out/pmc_branch_$WL.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "step" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s mean per launch %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
