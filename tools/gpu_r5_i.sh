#!/bin/bash
# round 5, device-built site plans: the byte-for-byte tests first, then the whole-job breakdown with the plan built by
# the device / by the host, the CLI's block path (record columns a segment at a time), the GPU suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_device.py -x -q > $O/pytest_plan_device.txt 2>&1; tail -15 $O/pytest_plan_device.txt
for wl in c4 c2x16 c10k; do
  for who in dev host; do
    timeout 600 python tools/e2e_breakdown.py $wl $who 2>&1 | grep -v amdgpu > $O/e2e_${wl}_$who.txt; echo "== $wl $who"; tail -5 $O/e2e_${wl}_$who.txt
  done
done
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu > $O/pytest_cli.txt 2>&1; tail -3 $O/pytest_cli.txt
timeout 1500 python tools/cli_block_time.py 10240 512 > $O/cli_block_time.txt 2>&1; cat $O/cli_block_time.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
