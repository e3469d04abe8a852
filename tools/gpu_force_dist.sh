#!/bin/bash
# The N-rank exchange path under the real RCCL backend with ONE rank (bench.py --force-dist),
# started through the launcher exactly as the driver starts an N-GPU run: the cost of the
# side-stream statistics / all-gather (c10k, c4, c3, c2x16) and of the particle filter's
# collectives (c5) against the plain pass -> gpurun_out/force_dist.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/force_dist.jsonl
for wl in ${@:-c10k c4 c3 c5 c2x16}; do
  steps=10; warm=2; [ "$wl" = c5 ] && steps=400 && warm=40   # (c5: a 0.2 ms cycle -- RCCL's first-collective costs need a real warm-up)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus 1 --force-dist --workload $wl --steps $steps --warmup $warm \
      --no-cpu-baseline --no-fill-probe > gpurun_out/force_dist_$wl.log 2>&1
  echo "rc=$? $wl"
  grep '^{' gpurun_out/force_dist_$wl.log | tail -1 >> gpurun_out/force_dist.jsonl
  tail -3 gpurun_out/force_dist_$wl.log | cut -c1-600
done
python3 - <<'PY'
import json
for line in open("gpurun_out/force_dist.jsonl"):
    j = json.loads(line)
    d = j["config"].get("dist_overhead", {})
    print(j["config"]["workload"][:40], "| dist %.3f plain %.3f overhead %.3f ms eff %.3f | gather_full %s" % (
        d.get("ms_per_step_dist", -1), d.get("ms_per_step_plain", -1), d.get("dist_overhead_ms", -1),
        d.get("predicted_weak_scaling_efficiency", -1), (j["config"].get("gather_full") or {}).get("ms")))
PY
