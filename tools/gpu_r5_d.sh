#!/bin/bash
# round 5, fourth GPU call: the whole GPU suite on the cleaned-up tree (probes behind SIPNET_PROBES, placement markers,
# ShardPool, the one-launch analysis with slots writing their own runs), c5's cycle and kernels, and the headline
# kernels against the round-4 library (the markers must not have moved anything)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5d
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.txt 2>&1
tail -1 $O/bench_c5.txt | cut -c1-400
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c5 -o c5 -- python3 $R/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-fill-probe --no-end-to-end > $R/$O/prof_c5.log 2>&1
cd $R
find $O/prof_c5 -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-150 | head -12
for wl in c10k c4 c3; do
  timeout 900 python tools/variant_bench.py --workload $wl --reps 5 r4base product >> $O/variant_headline.txt 2>&1
done
cat $O/variant_headline.txt
