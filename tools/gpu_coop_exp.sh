#!/bin/bash
# dev: time variants of the cooperative kernel with parts of the hand-over chain cut (COOP_EXP)
cd "$GRAFT_REPO_ROOT/sipnet_amd/csrc" || exit 1
cp ../libsipnet_amd.so /tmp/lib_orig.so
for e in ${EXPS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-gpu-rdc -DCOOP_EXP=$e -c step_coop.hip -o /tmp/step_coop_e.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipnet_amd.so engine.o step_kernel.o step_fast.o /tmp/step_coop_e.o pf.o plan.o host_io.o restart_io.o || exit 1
  (cd ../..; SIPNET_COOP=1 timeout 100 python bench.py --workload c10k --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('EXP $e: kernel_ms %.3f  cycles/step %.0f' % (j['roofline']['kernel_ms'], j['roofline']['kernel_ms']*1e-3*2.4e9/17520))")
done
cp /tmp/lib_orig.so ../libsipnet_amd.so
