#!/bin/bash
# round 4, first GPU call: the new node / peer-filter tests, then the c5 exchange path timed three ways
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_node.py tests/test_c_consumer.py tests/test_gpu_pf.py -m gpu -x -q > gpurun_out/r4a/pytest_node.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a/pytest_node.log
tail -5 gpurun_out/r4a/pytest_node.log
# C consumer at c5's shape
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sipnet_amd import synth
synth.write_clim('/tmp/day.clim', synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for dev in 0 0,0; do
  timeout 300 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim 131072 $dev 200 48 > gpurun_out/r4a/pf_consumer_$dev.log 2>&1
  echo "rc=$?" >> gpurun_out/r4a/pf_consumer_$dev.log
  cat gpurun_out/r4a/pf_consumer_$dev.log
done
# bench: force-dist c5 under torch.distributed.run (RCCL one rank), both exchanges
for ex in peer alltoall; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload c5 --force-dist --pf-exchange $ex --steps 200 --warmup 20 --no-cpu-baseline --no-fill-probe > gpurun_out/r4a/force_dist_c5_$ex.log 2>&1
  echo "rc=$?" >> gpurun_out/r4a/force_dist_c5_$ex.log
  grep -o '"dist_overhead": {[^}]*}' gpurun_out/r4a/force_dist_c5_$ex.log
done
# two processes on one GPU, IPC-mapped peers (gloo rehearsal)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --rehearse --workload c5 --members 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-fill-probe > gpurun_out/r4a/rehearse_c5_peer.log 2>&1
echo "rc=$?" >> gpurun_out/r4a/rehearse_c5_peer.log
tail -3 gpurun_out/r4a/rehearse_c5_peer.log | cut -c1-1500
