#!/bin/bash
# round 5, third GPU call: the one-launch particle-filter analysis (pfFusedKernel) against every filter test, the CLI
# block / --sites checkpoint tests, and c5's cycle with its kernels
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pf.py tests/test_gpu_node.py tests/test_gpu_multirank.py tests/test_c_consumer.py -x -q -m gpu > $O/pytest_pf.txt 2>&1
tail -8 $O/pytest_pf.txt
timeout 1500 python -m pytest tests/test_cli.py tests/test_gpu_restart.py -x -q -m gpu > $O/pytest_cli_restart.txt 2>&1
tail -15 $O/pytest_cli_restart.txt
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.txt 2>&1
tail -1 $O/bench_c5.txt | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_c5 -o c5 -- python3 $OLDPWD/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-fill-probe --no-end-to-end > $OLDPWD/$O/prof_c5.log 2>&1
cd $OLDPWD
find $O/prof_c5 -name '*kernel_stats.csv' | head -1 | xargs -r head -14
