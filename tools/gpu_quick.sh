#!/bin/bash
# quick loop: parity tests + a short bench sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle oracle 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -${LINES_T:-25}
for wl in ${WLS:-c2 c10k c3 c4}; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: print(l[:300]); continue
    print(j['config']['workload'][:40], '| %.3f G/s | %.2f ms | frac %.3f | dNEE %s' % (j['value']/1e9, j['ms_per_step'], j['roofline']['frac'], j['parity']))
"
done
