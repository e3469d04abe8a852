#!/bin/bash
mkdir -p gpurun_out/r4b
python -m pytest tests/test_c_consumer.py tests/test_gpu_pf.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r4b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4b/pytest.log
tail -3 gpurun_out/r4b/pytest.log
python tools/flag_sets_table.py gpurun_out/r4b/flag_sets_before.md > gpurun_out/r4b/flag_sets_before.log 2>&1
cat gpurun_out/r4b/flag_sets_before.log
