#!/bin/bash
# round 5, eighth GPU call: the whole GPU suite (diagnostics counters with the nitrogen cycle on the cooperative kernels,
# --bounded-waits from the CLI, ...), the headline kernels against the round-4 library once more, the CLI's block timing
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5h
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
for wl in c10k c4 c3 c10kn; do
  timeout 900 python tools/variant_bench.py --workload $wl --reps 5 r4base product >> $O/variant_headline.txt 2>&1
done
cat $O/variant_headline.txt
timeout 1500 python tools/cli_block_time.py 10240 512 > $O/cli_block_time.txt 2>&1
cat $O/cli_block_time.txt
