#!/bin/bash
# the whole GPU suite, then a FUZZ_OPT campaign
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4d
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4d/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4d/pytest_gpu.log
tail -15 gpurun_out/r4d/pytest_gpu.log
FUZZ_OPT=1 timeout 1500 python tools/fuzz_gpu.py ${1:-300} 4242 > gpurun_out/r4d/fuzz_opt.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/r4d/fuzz_opt.log
tail -4 gpurun_out/r4d/fuzz_opt.log
