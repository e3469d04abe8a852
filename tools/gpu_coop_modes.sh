#!/bin/bash
# dev: one-wave vs cooperative (LDS ring / HBM ring) kernels across the bench workloads
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in ${WLS:-c10k c4 c3}; do
  for mode in ${MODES:-0 1 2}; do
    SIPNET_COOP=$mode timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); print('$wl coop=$mode | %.3f G/s | kernel %.2f ms | dNEE %.2e' % (j['value']/1e9, j['roofline']['kernel_ms'], j['parity']['max_abs_dNEE']))
except Exception as e: print('$wl coop=$mode failed', e)"
  done
done
