#!/bin/bash
# timeline of one forcing handed to an idle device at c4: kernels and copies (rocprofv3 --kernel-trace --memory-copy-trace)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -- python3 tools/e2e_breakdown.py c4 > $O/e2e.txt 2>&1
cp $(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
cp $(find /tmp/prof_tl -name "*memory_copy_trace.csv" | head -1) $O/memory_copy_trace.csv
tail -4 $O/e2e.txt | cut -c1-200
