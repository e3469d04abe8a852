#!/bin/bash
# round 5, seventh GPU call: the new cooperative instantiations (four-chunk optional physics in fp32, "everything" + record),
# the two-level barrier of the one-launch analysis, flag-set tables, the CLI's block against its text
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5g
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_configs.py tests/test_gpu_full.py tests/test_gpu_flags.py tests/test_gpu_pf.py tests/test_gpu_node.py tests/test_gpu_multirank.py tests/test_c_consumer.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
SIPNET_LIB=build/variants/pfstamps/libsipnet_amd.so timeout 300 python tools/pf_analysis_time.py 131072 200 2>&1 | grep 'ms per\|phases' > $O/pf_stamps.txt
timeout 300 python tools/pf_analysis_time.py 131072 200 2>&1 | grep 'ms per' >> $O/pf_stamps.txt
cat $O/pf_stamps.txt
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.txt 2>&1
tail -1 $O/bench_c5.txt | cut -c1-260
timeout 1500 python tools/flag_sets_table.py $O/flag_sets.md c10k c4 c3f32 c10krec > $O/flag_sets.log 2>&1
cat $O/flag_sets.md
timeout 1500 python tools/cli_block_time.py 10240 512 > $O/cli_block_time.txt 2>&1
cat $O/cli_block_time.txt
