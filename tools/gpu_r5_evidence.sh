#!/bin/bash
# round 5 evidence on the current binary: rocprofv3 profiles of every workload (kernel trace + separate PMC passes ->
# gpurun_out/prof_r05_*), the bench line of every workload, the N-rank exchange path under one-rank RCCL, the particle
# filter's cycle from C, instruction mixes of c3 / c5's kernels, the whole-job breakdown, the CLI's block against its text
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
O=gpurun_out/r5ev; mkdir -p $O
bash tools/gpu_profile_all.sh r05 c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3 > $O/profile_all.log 2>&1
: > $O/bench_all.jsonl
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do
  extra="--no-cpu-baseline"; [ "$wl" = c10k ] && extra=""
  steps=5; [ "$wl" = c5 ] && steps=200
  timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 $extra 2>$O/bench_$wl.err | grep '^{' | tail -1 >> $O/bench_all.jsonl
  echo "bench $wl rc=$?"
done
bash tools/gpu_force_dist.sh c10k c2x16 c4 c3 c5 c10kn c4n c10kr3 > $O/force_dist.log 2>&1
cp gpurun_out/force_dist.jsonl $O/force_dist.jsonl
tail -9 $O/force_dist.log
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sipnet_amd import synth
synth.write_clim('/tmp/day.clim', synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for dev in 0 0,0; do
  timeout 300 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim 131072 $dev 300 48 > $O/pf_consumer_$dev.log 2>&1
  echo "rc=$?" >> $O/pf_consumer_$dev.log
done
cat $O/pf_consumer_0.log
for wl in c3 c5 c4; do bash tools/gpu_pmc_branch.sh $wl > /dev/null 2>&1; cp gpurun_out/pmc_branch_$wl.txt $O/; done
for wl in c4 c2x16 c10k; do for who in dev host; do timeout 600 python tools/e2e_breakdown.py $wl $who 2>&1 | grep -v amdgpu > $O/e2e_${wl}_$who.txt; tail -3 $O/e2e_${wl}_$who.txt | cut -c1-200; done; done
timeout 300 python tools/plan_device_time.py 2>&1 | grep -v amdgpu > $O/plan_device_time.txt
rm -rf /tmp/prof_plan_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_plan_c4 -- python3 tools/e2e_breakdown.py c4 > $O/e2e_prof_c4.txt 2>&1
cp "$(find /tmp/prof_plan_c4 -name '*kernel_stats.csv' | head -1)" $O/plan_kernel_stats_c4.csv
timeout 1500 python tools/cli_block_time.py 10240 512 > $O/cli_block_time.txt 2>&1
cat $O/cli_block_time.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
