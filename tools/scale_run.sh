#!/bin/bash
# The first run on a node with more than one GPU, as ONE command: the bench line of c10k, c4 and c5 at N = 1, 2, 4, 8
# through torch.distributed.run (one process per GPU over RCCL: what the driver's SCALE run does), and the same three
# shapes through the C host (sipnet_node_*: one process, one thread + one RCCL rank per GPU, the all-gather issued from C;
# tests/c/node_consumer.c, tests/c/pf_consumer.c).  Everything lands under gpurun_out/scale/ (copy what is to be
# judged into profiles/).  usage: tools/scale_run.sh [max_gpus]   (default: every visible GPU, at most 8)
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
MAX=${1:-$HAVE}; [ "$MAX" -gt 8 ] && MAX=8
O=gpurun_out/scale; mkdir -p $O
echo "visible GPUs: $HAVE, running up to $MAX" | tee $O/README.txt
# 1. the two-real-GPU tests that one-GPU boxes skip (member and site shards, the overlapped gather, peer-read filter cycles)
if [ "$HAVE" -ge 2 ]; then
  timeout 1800 python3 -m pytest tests/test_gpu_node.py tests/test_gpu_multirank.py -q -m gpu -k "two_real or real_devices" > $O/pytest_two_gpus.txt 2>&1
  tail -3 $O/pytest_two_gpus.txt
fi
# 2. one process per GPU (torch.distributed.run, RCCL): the bench contract's lines
PORT=29551
for wl in c10k c4 c5; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$MAX" ] && continue
    if [ "$n" -eq 1 ]; then
      timeout 1200 python3 bench.py --workload $wl --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_${wl}_n1.log 2>&1
    else
      PORT=$((PORT + 1))
      timeout 1200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
          bench.py --workload $wl --gpus $n --steps 20 --warmup 3 > $O/bench_${wl}_n$n.log 2>&1
    fi
    grep '^{' $O/bench_${wl}_n$n.log | tail -1 >> $O/scale_lines.jsonl
    grep '^{' $O/bench_${wl}_n$n.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', 'N=%d' % d['n_gpus'], 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'])"
  done
done
# 3. the C host: one process, one shard per GPU (sipnet_node_*), statistics all-gather and overlapped plane gather from C
gcc -std=c99 -O1 -Iinclude tests/c/node_consumer.c -o /tmp/node_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from sipnet_amd import synth
synth.write_clim("/tmp/year.clim", synth.round_like_file(synth.half_hourly_year_raw(17520)))
synth.write_clim("/tmp/day.clim", synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for n in 1 2 4 8; do
  [ "$n" -gt "$MAX" ] && continue
  devs=$(seq -s, 0 $((n - 1)))
  # weak scaling: 10 240 members / 131 072 particles PER GPU
  timeout 900 /tmp/node_consumer sipnet_amd/data/base_forest.param /tmp/year.clim $((10240 * n)) $devs > $O/node_consumer_n$n.log 2>&1
  echo "rc=$?" >> $O/node_consumer_n$n.log
  timeout 900 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim $((131072 * n)) $devs 300 48 > $O/pf_consumer_n$n.log 2>&1
  echo "rc=$?" >> $O/pf_consumer_n$n.log
  grep -h 'ms\|rc=' $O/node_consumer_n$n.log $O/pf_consumer_n$n.log | head -12
done
ls $O
