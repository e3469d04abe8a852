#!/bin/bash
# the bench line of every workload on the current binary and the committed counters -> gpurun_out/bench_all.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/bench_all.jsonl
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do
  extra="--no-cpu-baseline"; [ "$wl" = c10k ] && extra=""
  steps=5; [ "$wl" = c5 ] && steps=200
  timeout 900 python bench.py --workload $wl --steps $steps --warmup 2 $extra 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/bench_all.jsonl
  echo "bench $wl rc=$?"
done
