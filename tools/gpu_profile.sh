#!/bin/bash
# rocprofv3 evidence for one workload: kernel-trace stats of the bench command, then
# PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes)
WL=${1:-c10k}; TAG=${2:-r02}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_${WL}; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-fill-probe --no-end-to-end > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_target.py $WL 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_target.py $WL 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 tools/prof_target.py $WL 2 > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -name "*counter_collection.csv"); do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d mean=%.6g" % (len(v), sum(v) / len(v)))
PY
done
tail -2 $OUT/pmc_fetch.log
