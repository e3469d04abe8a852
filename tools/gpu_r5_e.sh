#!/bin/bash
# round 5, fifth GPU call: the parameter index of a one-batch filter, "everything" + record on the cooperative kernels,
# and where the one-launch analysis spends its 40 us (barrier variants)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pf.py tests/test_gpu_full.py tests/test_gpu_node.py tests/test_gpu_flags.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for v in "" pfb128 pfs32 pfb128s32 pfb256s16; do
  lib=""; [ -n "$v" ] && lib=build/variants/$v/libsipnet_amd.so
  SIPNET_LIB=$lib timeout 300 python tools/pf_analysis_time.py 131072 200 2>&1 | grep 'ms per' >> $O/pf_analysis_time.txt
done
cat $O/pf_analysis_time.txt
timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5.txt 2>&1
tail -1 $O/bench_c5.txt | cut -c1-300
