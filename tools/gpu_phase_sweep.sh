#!/bin/bash
# the code-phase sweep (coopBody's prologue: .p2align 5 + k s_nop): every library under build/variants (ph0..ph7 from
# tools/build_variants.py ph$k=-DSIPNET_PAD_NOPS=$k) on the instantiations the workloads launch
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/phase_sweep.txt; : > $out
run() { echo "== $*" | tee -a $out; timeout 900 env $1 python tools/variant_bench.py --reps 3 "${@:2}" 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    n = l.split()[0]; j = l[l.index('{'):] if '{' in l else None
    if j:
        d = json.loads(j); print('%-8s %-52s min %.4f med %.4f dNEE %.1e' % (n, d['kernel'], d['ms_min'], d['ms_med'], d['dNEE']))
    else: print(l.strip()[:200])
" | tee -a $out; }
if [ "$SWEEP_SHORT" = 3 ]; then   # the kernels with a soil / factor wave
  run X=0 --workload c10kn; run X=0 --workload c4n; run X=0 --workload c10k
  exit 0
fi
if [ -n "$SWEEP_SHORT" ]; then   # the seven instantiations of the bench workloads + the full-state headline build
  run X=0 --workload c10k; run X=0 --workload c4; run X=0 --workload c10kn
  run X=0 --workload c10kr3; run X=0 --workload c10k --kopt 4
  [ "$SWEEP_SHORT" = 2 ] && { run X=0 --workload c3; run X=0 --workload c4n; }
  exit 0
fi
run X=0 --workload c10k
run X=0 --workload c4
run X=0 --workload c3
run X=0 --workload c10kn
run X=0 --workload c4n
run X=0 --workload c10kr3
run VB_PREC=f32 --workload c10k
run VB_PREC=f32 --workload c4
run X=0 --workload c10k --kernel coop_hbm
run X=0 --workload c10k --kopt 4
run X=0 --workload c4 --kopt 4
run VB_PREC=f64 --workload c3
