#!/bin/bash
# kernel trace of the whole-job hand-over with device-built plans (how long do the four plan kernels take?)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
for wl in c4 c10k; do
  rm -rf /tmp/prof_plan_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_plan_$wl -- python3 tools/e2e_breakdown.py $wl > $O/e2e_prof_$wl.txt 2>&1
  f=$(find /tmp/prof_plan_$wl -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/plan_kernel_stats_$wl.csv
  echo "== $wl"; grep -i "plan\|Name" $O/plan_kernel_stats_$wl.csv | cut -c1-200
done
