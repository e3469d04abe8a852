#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 0 1 2 4 3 7; do
  echo -n "dbg=$d: "
  SIPNET_DBG=$d python bench.py --workload ${1:-c10k} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('%.2f ms kernel %.2f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
done
