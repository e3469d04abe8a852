#!/bin/bash
# device-built site plans, second pass: tests, the walk per forcing shape, whole-job breakdown device / host, kernel trace
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_device.py -x -q > $O/pytest_plan_device.txt 2>&1; tail -12 $O/pytest_plan_device.txt
timeout 300 python tools/plan_device_time.py 2>&1 | grep -v amdgpu > $O/plan_device_time.txt; cat $O/plan_device_time.txt
for wl in c4 c2x16 c10k; do
  for who in dev host; do
    timeout 600 python tools/e2e_breakdown.py $wl $who 2>&1 | grep -v amdgpu > $O/e2e_${wl}_$who.txt; echo "== $wl $who"; tail -6 $O/e2e_${wl}_$who.txt | cut -c1-330
  done
done
rm -rf /tmp/prof_plan_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_plan_c4 -- python3 tools/e2e_breakdown.py c4 > $O/e2e_prof_c4.txt 2>&1
cp "$(find /tmp/prof_plan_c4 -name '*kernel_stats.csv' | head -1)" $O/plan_kernel_stats_c4.csv
grep -i "plan\|Name" $O/plan_kernel_stats_c4.csv | cut -c1-200
