#!/bin/bash
# round 4: the optional-physics cooperative kernels -- parity tests, then the flag-set table
mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests/test_gpu_flags.py -m gpu -x -q > gpurun_out/r4c/pytest_flags.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4c/pytest_flags.log
tail -25 gpurun_out/r4c/pytest_flags.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "x_ or coop_ncycle" > gpurun_out/r4c/pytest_configs_x.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4c/pytest_configs_x.log
tail -15 gpurun_out/r4c/pytest_configs_x.log
timeout 900 python tools/flag_sets_table.py gpurun_out/r4c/flag_sets_after.md > gpurun_out/r4c/flag_sets_after.log 2>&1
cat gpurun_out/r4c/flag_sets_after.log
