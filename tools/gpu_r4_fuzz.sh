#!/bin/bash
# round 4 fuzz campaigns on the final binary (bounded-wait build of the cooperative kernels: a hang would be a report)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4fz
export FUZZ_BOUNDED=1
FUZZ_OPT=1 timeout 1500 python tools/fuzz_gpu.py ${1:-3000} 90401 > gpurun_out/r4fz/opt.log 2>&1; echo "rc=$?" >> gpurun_out/r4fz/opt.log; tail -2 gpurun_out/r4fz/opt.log
FUZZ_NCYC=1 timeout 900 python tools/fuzz_gpu.py ${2:-1500} 90402 > gpurun_out/r4fz/ncyc.log 2>&1; echo "rc=$?" >> gpurun_out/r4fz/ncyc.log; tail -2 gpurun_out/r4fz/ncyc.log
FUZZ_COOP=1 timeout 900 python tools/fuzz_gpu.py ${3:-1500} 90403 > gpurun_out/r4fz/coop.log 2>&1; echo "rc=$?" >> gpurun_out/r4fz/coop.log; tail -2 gpurun_out/r4fz/coop.log
unset FUZZ_BOUNDED
timeout 1500 python tools/fuzz_gpu.py ${4:-3000} 90404 > gpurun_out/r4fz/all.log 2>&1; echo "rc=$?" >> gpurun_out/r4fz/all.log; tail -2 gpurun_out/r4fz/all.log
# sites of different lengths in one batch, on top of each campaign (bounded-wait build)
export FUZZ_BOUNDED=1 FUZZ_RAGGED=1
for c in FUZZ_COOP FUZZ_OPT FUZZ_NCYC FUZZ_NONE; do
  env $c=1 timeout 900 python tools/fuzz_gpu.py ${5:-500} 90405 > gpurun_out/r4fz/ragged_$c.log 2>&1; echo "rc=$?" >> gpurun_out/r4fz/ragged_$c.log; tail -2 gpurun_out/r4fz/ragged_$c.log
done
