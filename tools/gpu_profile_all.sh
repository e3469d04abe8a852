#!/bin/bash
# rocprofv3 evidence for every bench workload (tools/gpu_profile.sh each): kernel-trace stats of the
# bench command + separate PMC passes -> gpurun_out/prof_<tag>_<workload>/; distil on the CPU side with
#   for w in ...; do python tools/distill_profile.py gpurun_out/prof_<tag>_$w; done
TAG=${1:-r03}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for wl in ${@:-c10k c2 c2x16 c3 c4 c5 c10kn}; do
  echo "==== $wl"
  bash tools/gpu_profile.sh $wl $TAG > gpurun_out/prof_${TAG}_${wl}.log 2>&1
  tail -2 gpurun_out/prof_${TAG}_${wl}.log
done
