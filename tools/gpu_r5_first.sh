#!/bin/bash
# round 5, first GPU call: narrow record fields of fp32-mixed batches (plan.h) -- parity of the fp32 paths, c3 / c5
# against the round-4 library, and what a day / night step costs on the layouts a two-members-per-lane kernel
# would have to use at c3's size (NOTES "Round 5: packed fp32")
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_f32.txt 2>&1
tail -5 $O/pytest_f32.txt
for lib in r4base product; do
  timeout 600 python tools/variant_bench.py --workload c3 --reps 5 $lib >> $O/variant_c3.txt 2>&1
done
cat $O/variant_c3.txt
for lib in build/variants/r4base/libsipnet_amd.so ""; do
  echo "== lib: ${lib:-product}" >> $O/day_night_f32.txt
  SIPNET_LIB=$lib M=65536 PREC=f32 timeout 600 python tools/day_night_time.py coop_quad >> $O/day_night_f32.txt 2>&1
  SIPNET_LIB=$lib M=32768 PREC=f32 timeout 600 python tools/day_night_time.py coop_pair >> $O/day_night_f32.txt 2>&1
  SIPNET_LIB=$lib M=16384 PREC=f32 timeout 600 python tools/day_night_time.py coop_lds >> $O/day_night_f32.txt 2>&1
  SIPNET_LIB=$lib M=131072 PREC=f32 timeout 600 python tools/day_night_time.py one_wave >> $O/day_night_f32.txt 2>&1
done
cat $O/day_night_f32.txt
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.txt 2>&1
tail -1 $O/bench_c5.txt | cut -c1-600
