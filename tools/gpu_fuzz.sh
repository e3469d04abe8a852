#!/bin/bash
# the fuzz campaigns on the final binary (round 5 on): the new instantiations first (FUZZ_R5: optional physics at four chunks per
# workgroup in fp32-mixed batches, records + diagnostics counters from the optional-physics and nitrogen-cycle kernels),
# then round 4's campaigns again (every fp32-mixed trial now runs on narrow record fields); SEED_OFFSET=k shifts every campaign's seed
cd "$(dirname "$0")/.." || exit 1
O=${FUZZ_OUT:-gpurun_out/fuzz}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 1500 python tools/fuzz_gpu.py $N $SEED > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log; echo "== $name: $(grep -c '^trial' $O/$name.log) trials, $(tail -1 $O/$name.log)"; grep -i 'mismatch\|error\|assert' $O/$name.log | head -3; }
N=${1:-2000} SEED=$((100501 + ${SEED_OFFSET:-0})) run r5_opt FUZZ_R5=1 FUZZ_OPT=1
N=${2:-1200} SEED=$((100502 + ${SEED_OFFSET:-0})) run r5_ncyc FUZZ_R5=1 FUZZ_NCYC=1
N=${3:-1000} SEED=$((100503 + ${SEED_OFFSET:-0})) run coop_bounded FUZZ_COOP=1 FUZZ_BOUNDED=1
N=${4:-2000} SEED=$((100504 + ${SEED_OFFSET:-0})) run all FUZZ_NONE=1
N=${5:-400} SEED=$((100505 + ${SEED_OFFSET:-0})) run ragged_opt FUZZ_R5=1 FUZZ_OPT=1 FUZZ_RAGGED=1
N=${5:-400} SEED=$((100506 + ${SEED_OFFSET:-0})) run ragged_ncyc FUZZ_R5=1 FUZZ_NCYC=1 FUZZ_RAGGED=1
N=${5:-400} SEED=$((100507 + ${SEED_OFFSET:-0})) run ragged_coop FUZZ_COOP=1 FUZZ_BOUNDED=1 FUZZ_RAGGED=1
# round 6: the in-kernel sums (sipnet_batch_run_sums) on the cooperative layouts, bit for bit against the trial's own planes
N=${6:-600} SEED=$((100608 + ${SEED_OFFSET:-0})) run coop_sums FUZZ_COOP=1 FUZZ_SUMS=1
N=${7:-300} SEED=$((100609 + ${SEED_OFFSET:-0})) run ragged_coop_sums FUZZ_COOP=1 FUZZ_SUMS=1 FUZZ_RAGGED=1
N=${8:-400} SEED=$((100610 + ${SEED_OFFSET:-0})) run opt_sums FUZZ_R5=1 FUZZ_OPT=1 FUZZ_SUMS=1
N=${9:-400} SEED=$((100611 + ${SEED_OFFSET:-0})) run ncyc_sums FUZZ_R5=1 FUZZ_NCYC=1 FUZZ_SUMS=1
# ... and over the general generator (one-wave kernel included: step_fast_sums.hip, held to the arithmetic's last bits)
N=${10:-500} SEED=$((100812 + ${SEED_OFFSET:-0})) run all_sums FUZZ_NONE=1 FUZZ_SUMS=1
