#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for wl in c2 c10k c3 c4; do
  for fm in 0 1; do
    [ "$wl" = c3 ] && [ "$fm" = 0 ] && continue
    timeout 600 python bench.py --workload $wl --fast-math $fm --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/sweep.jsonl
  done
done
