#!/bin/bash
# round 4 evidence on the current binary: rocprofv3 profiles of every workload (gpurun_out/prof_r04_*), the bench lines of
# all workloads, the N-rank exchange path under one-rank RCCL for all of them, the particle filter's cycle from C, the
# full-record kernels of the optional flag sets
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
mkdir -p gpurun_out/r4ev
bash tools/gpu_profile_all.sh r04 c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3 > gpurun_out/r4ev/profile_all.log 2>&1
: > gpurun_out/r4ev/bench_all.jsonl
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do
  extra="--no-cpu-baseline"; [ "$wl" = c10k ] && extra=""
  steps=5; [ "$wl" = c5 ] && steps=200
  timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 $extra 2>gpurun_out/r4ev/bench_$wl.err | grep '^{' | tail -1 >> gpurun_out/r4ev/bench_all.jsonl
  echo "bench $wl rc=$?"
done
bash tools/gpu_force_dist.sh c10k c2x16 c4 c3 c5 c10kn c4n c10kr3 > gpurun_out/r4ev/force_dist.log 2>&1
cp gpurun_out/force_dist.jsonl gpurun_out/r4ev/force_dist.jsonl
tail -9 gpurun_out/r4ev/force_dist.log
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sipnet_amd import synth
synth.write_clim('/tmp/day.clim', synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for dev in 0 0,0; do
  timeout 300 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim 131072 $dev 300 48 > gpurun_out/r4ev/pf_consumer_$dev.log 2>&1
  echo "rc=$?" >> gpurun_out/r4ev/pf_consumer_$dev.log
done
cat gpurun_out/r4ev/pf_consumer_0.log
for f in default russell_2 russell_3; do python tools/full_record_time.py 10240 $f 2>&1 | grep -v amdgpu; done > gpurun_out/r4ev/full_record_flag_sets.txt
cat gpurun_out/r4ev/full_record_flag_sets.txt
