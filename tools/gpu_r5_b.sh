#!/bin/bash
# round 5, second GPU call: the ensemble output block and --sites checkpoints through the CLI; wave priorities by role
# (build/variants/prio*: -DSIPNET_PROBE_PRIO) on the layouts whose waves share a SIMD (c3 quad, c4 pair)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5b
mkdir -p $O
timeout 1500 python -m pytest tests/test_cli.py tests/test_gpu_restart.py -x -q -m gpu > $O/pytest_cli_restart.txt 2>&1
tail -15 $O/pytest_cli_restart.txt
for wl in c3 c4; do
  timeout 900 python tools/variant_bench.py --workload $wl --reps 5 product prioC prioCWL prioLCW prioCL >> $O/variant_prio.txt 2>&1
done
cat $O/variant_prio.txt
