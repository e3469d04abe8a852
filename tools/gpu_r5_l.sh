#!/bin/bash
# device-built plans, third pass: the GPU suite, whole-job numbers (tool + bench lines with end_to_end / pipelined)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
for wl in c4 c2x16; do
  timeout 600 python tools/e2e_breakdown.py $wl dev 2>&1 | grep -v amdgpu > $O/e2e_${wl}_dev.txt; echo "== $wl dev"; tail -5 $O/e2e_${wl}_dev.txt | cut -c1-330
done
: > $O/bench.jsonl
for wl in c4 c2x16 c10k c3; do
  timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>$O/bench_$wl.err | grep '^{' | tail -1 >> $O/bench.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r5l/bench.jsonl'):
    d = json.loads(l); e = d['roofline'].get('end_to_end') or d.get('end_to_end')
    print(d['config']['workload'][:28], 'ms/step', round(d['ms_per_step'], 3), 'e2e', e and {k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items() if k in ('ms', 'pipelined_ms', 'plan_device_sites')})
PY
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
