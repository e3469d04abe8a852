#!/bin/bash
# second SQ counter pass: where a step's cycles go by instruction class
WL=${1:-c10k}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc2_$WL; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH --output-format csv -d $OUT/a -- python3 tools/prof_target.py $WL 2 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/b -- python3 tools/prof_target.py $WL 2 > $OUT/b.log 2>&1
for f in $(find $OUT -name "*counter_collection.csv"); do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "stepFast" in r["Kernel_Name"] or "stepCoop" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "mean=%.6g" % (sum(v) / len(v)))
PY
done
