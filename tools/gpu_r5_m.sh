#!/bin/bash
# the final bench lines of all workloads (traffic from the
# round-5 profiles), the GPU suite
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5m; mkdir -p $O
: > $O/bench_all.jsonl
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do
  extra="--no-cpu-baseline"; [ "$wl" = c10k ] && extra=""
  steps=5; [ "$wl" = c5 ] && steps=200
  timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 $extra 2>$O/bench_$wl.err | grep '^{' | tail -1 >> $O/bench_all.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5m/bench_all.jsonl'):
    d = json.loads(l); r = d['roofline']; e = r.get('end_to_end') or {}
    print(d['config']['workload'][:30], 'ms', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 3), 'tag', str(r.get('traffic_tag'))[:12], 'e2e', e.get('ms') and round(e['ms'], 2), e.get('pipelined_ms') and round(e['pipelined_ms'], 2))
PY
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sipnet_amd import synth
synth.write_clim('/tmp/day.clim', synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for dev in 0 0,0; do timeout 300 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim 131072 $dev 300 48 > $O/pf_consumer_$dev.log 2>&1; echo "rc=$?" >> $O/pf_consumer_$dev.log; grep "ms_per\|identical\|rc=" $O/pf_consumer_$dev.log; done
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
