#!/bin/bash
# the last GPU call of round 5: c5's profile on the final binary, the bench lines of all workloads
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD} TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5n; mkdir -p $O
bash tools/gpu_profile.sh c5 r05 > $O/prof_c5.log 2>&1; tail -2 $O/prof_c5.log
: > $O/bench_all.jsonl
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do
  extra="--no-cpu-baseline"; [ "$wl" = c10k ] && extra=""
  steps=5; [ "$wl" = c5 ] && steps=200
  timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 $extra 2>$O/bench_$wl.err | grep '^{' | tail -1 >> $O/bench_all.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5n/bench_all.jsonl'):
    d = json.loads(l); r = d['roofline']; e = r.get('end_to_end') or {}
    print(d['config']['workload'][:30], 'ms', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 3), 'tag', str(r.get('traffic_tag'))[:12], 'e2e', e.get('ms') and round(e['ms'], 2), e.get('pipelined_ms') and round(e['pipelined_ms'], 2))
PY
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
