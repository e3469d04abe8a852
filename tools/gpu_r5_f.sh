#!/bin/bash
# round 5, sixth GPU call: phase stamps of the one-launch analysis; where the whole-job time goes at c4 / c2x16
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5f
mkdir -p $O
SIPNET_LIB=build/variants/pfstamps/libsipnet_amd.so timeout 300 python tools/pf_analysis_time.py 131072 200 2>&1 | grep 'ms per\|phases' > $O/pf_stamps.txt
SIPNET_LIB=build/variants/pfstamps/libsipnet_amd.so timeout 300 python tools/pf_analysis_time.py 131072 200 2>&1 | grep 'phases' >> $O/pf_stamps.txt
cat $O/pf_stamps.txt
for wl in c4 c2x16; do
  timeout 600 python tools/e2e_breakdown.py $wl > $O/e2e_$wl.txt 2>&1
  tail -12 $O/e2e_$wl.txt
done
