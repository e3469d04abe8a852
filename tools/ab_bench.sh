#!/bin/bash
# A/B of two trees on ONE box: the kernel time of bench.py's workloads, alternating between the product tree and a
# worktree of an older commit built under build/<name> (git worktree add build/<name> <commit>; make -C .../csrc)
# usage: tools/ab_bench.sh <name> <workload>...
cd "$GRAFT_REPO_ROOT" || exit 1
name=$1; shift
for wl in "$@"; do
  for rep in 1 2 3; do
    for tree in . build/$name; do
      (cd $tree && timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-fill-probe --no-end-to-end 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$wl', '$tree', 'ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']), d['roofline']['kernel'])
")
    done
  done
done
