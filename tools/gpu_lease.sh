#!/bin/bash
# ONE script for every GPU lease (replaces the per-lease gpu_r4_* / gpu_r5_* scripts of earlier rounds):
#   gpurun --timeout S -- 'tools/gpu_lease.sh <tag> <recipe> [<recipe> ...]'
# Results land under gpurun_out/<tag>/ (copy what is to be judged into profiles/).  Recipes:
#   suite          the whole GPU test suite (pytest -m gpu)
#   tests:<expr>   pytest -m gpu -k <expr>
#   bench          the driver's default bench line
#   bench_all      one bench line per workload (no CPU baseline) -> bench_all.jsonl
#   force_dist     bench.py --force-dist for every workload + c5 at --pretend-world 8 -> force_dist.jsonl
#   pf             the cross-rank analysis at the slot counts of 1 / 2 / 4 / 8 ranks (tools/pf_peers_time.py), plain and
#                  under rocprofv3 --kernel-trace --stats
#   profile:<wl>   rocprofv3 --kernel-trace --stats of `bench.py --workload <wl>` -> <wl>_kernel_stats.csv
#   smoke          __graft_entry__.smoke()
#   node_gather    tools/node_gather_time.py at c10k's shape: plain run, planes gathered afterwards / overlapped, daily sums, fp32
#   pf_consumer    tests/c/pf_consumer.c at c5's shape (131 072 particles x 48 steps, 300 cycles): devices 0 and 0,0
#   evidence:<tag> the round's evidence on the current binary: rocprofv3 --kernel-trace --stats + separate --pmc passes of every
#                  workload (tools/gpu_profile_all.sh <tag>), then bench_all force_dist pf node_gather pf_consumer, instruction
#                  mixes of c3 / c4 / c5's kernels, the whole-job breakdown; distil on the CPU side with tools/collect_evidence.sh
#   evidence_rest  the same without the rocprofv3 passes (run those first -- tools/gpu_profile_all.sh <tag> -- and distil them, so
#                  that the bench lines carry the fresh counters' tag)
#   sums_time      tools/sums_time.py: the planes' launch against the in-launch sums at six shapes
#   fuzz_sums      the four sums campaigns of tools/gpu_fuzz.sh alone
#   fuzz           tools/gpu_fuzz.sh (the differential campaigns)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for R in "$@"; do
  case $R in
    suite)
      timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt ;;
    tests:*)
      timeout 1800 python3 -m pytest tests -q -m gpu -k "${R#tests:}" 2>&1 | tail -30 > $O/pytest_k.txt; tail -8 $O/pytest_k.txt ;;
    smoke)
      timeout 600 python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt ;;
    bench)
      timeout 1200 python3 bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
      python3 -c "import json; d=json.load(open('$O/bench_default.json')); print('value %.4g %s, ms/step %.3f, roofline %s' % (d['value'], d['unit'], d['ms_per_step'], {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic')}))" ;;
    bench_all)
      : > $O/bench_all.jsonl
      for wl in c10k c2 c2x16 c3 c4 c5 c10kn c10kr3 c4n; do
        steps=10; warm=2; [ "$wl" = c5 ] && steps=400 && warm=40
        timeout 900 python3 bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline > $O/bench_$wl.log 2>&1
        grep '^{' $O/bench_$wl.log | tail -1 >> $O/bench_all.jsonl
        grep '^{' $O/bench_$wl.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['roofline'].get('kernel', '')[:60])"
      done ;;
    force_dist)
      : > $O/force_dist.jsonl
      for wl in c10k c10ksums c4 c4sums c3 c3sums c5 c5p8 c5torch c5p8torch c2x16 c2x16sums c10kn c10knsums c4n c4nsums c10kr3 c10kr3sums; do
        steps=10; warm=2; extra=""; w=$wl
        [ "$wl" = c10ksums ] && w=c10k && extra="--gather sums"   # (every member's daily sums from the kernel's own launch, gathered under the next pass)
        [ "$wl" = c4sums ] && w=c4 && extra="--gather sums"
        [ "$wl" = c3sums ] && w=c3 && extra="--gather sums"
        case $wl in c2x16sums|c10knsums|c4nsums|c10kr3sums) w=${wl%sums}; extra="--gather sums" ;; esac
        [ "$wl" = c5 ] && steps=400 && warm=40   # (a 0.15 ms cycle: RCCL's first-collective costs need a real warm-up)
        [ "$wl" = c5p8 ] && steps=400 && warm=40 && extra="--pretend-world 8" && w=c5
        # (the filter's all-gather through torch.distributed's process group instead of the engine's own communicator)
        [ "$wl" = c5torch ] && steps=400 && warm=40 && extra="--pf-collective torch" && w=c5
        [ "$wl" = c5p8torch ] && steps=400 && warm=40 && extra="--pretend-world 8 --pf-collective torch" && w=c5
        timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
            bench.py --gpus 1 --force-dist --workload $w $extra --steps $steps --warmup $warm --no-cpu-baseline --no-fill-probe > $O/force_dist_$wl.log 2>&1
        echo "rc=$? $wl"
        grep '^{' $O/force_dist_$wl.log | tail -1 >> $O/force_dist.jsonl
      done
      python3 - "$O/force_dist.jsonl" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    j = json.loads(line)
    d = j["config"].get("dist_overhead", {})
    pf = j["config"].get("particle_filter", {})
    print(j["config"]["workload"][:30], "pretend", pf.get("pretend_world"), "| dist %.4f plain %.4f overhead %.4f ms eff %.3f | gather_full %s | crossing/cycle %s slots %s" % (
        d.get("ms_per_step_dist", -1), d.get("ms_per_step_plain", -1), d.get("dist_overhead_ms", -1),
        d.get("predicted_weak_scaling_efficiency", -1), (j["config"].get("gather_full") or {}).get("ms"), pf.get("crossing_per_cycle"), pf.get("analysis_slots")),
          "| gather", j["config"].get("gather"), "gather_sums", {k: v for k, v in (j["config"].get("gather_sums") or {}).items() if k in ("ms", "max_abs_diff_vs_planes", "kernel")})
PY
      ;;
    pf)
      timeout 900 python3 tools/pf_peers_time.py 131072 300 > $O/pf_peers_time.txt 2>&1; cat $O/pf_peers_time.txt
      rm -rf /tmp/prof_pf
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf -- python3 tools/pf_peers_time.py 131072 100 8 > $O/pf_peers_prof.txt 2>&1
      cp $(find /tmp/prof_pf -name "*kernel_stats.csv" | head -1) $O/pf_peers_w8_kernel_stats.csv 2>/dev/null
      head -12 $O/pf_peers_w8_kernel_stats.csv | cut -c1-160 ;;
    profile:*)
      wl=${R#profile:}; rm -rf /tmp/prof_$wl
      steps=10; warm=2; [ "$wl" = c5 ] && steps=400 && warm=40
      timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python3 bench.py --workload $wl --steps $steps --warmup $warm --no-cpu-baseline --no-fill-probe --no-end-to-end > $O/profile_$wl.log 2>&1
      cp $(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1) $O/${wl}_kernel_stats.csv 2>/dev/null
      head -6 $O/${wl}_kernel_stats.csv | cut -c1-200 ;;
    node_gather)
      for d in 0 0,0; do timeout 600 python3 tools/node_gather_time.py 10240 1 $d 2>&1 | grep -v amdgpu.ids > $O/node_gather_time_$d.txt; cat $O/node_gather_time_$d.txt; done ;;
    pf_consumer)
      gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
      python3 -c "import sys; sys.path.insert(0, '.'); from sipnet_amd import synth; synth.write_clim('/tmp/day.clim', synth.round_like_file(synth.half_hourly_year_raw(48)))"
      for dev in 0 0,0; do
        for ow in 0 1; do
          timeout 300 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim 131072 $dev 300 48 $ow > $O/pf_consumer_${dev}_onewave$ow.log 2>&1
          echo "rc=$?" >> $O/pf_consumer_${dev}_onewave$ow.log
        done
      done
      grep -h "ms_per_cycle\|state_identical\|kernel" $O/pf_consumer_*.log ;;
    evidence:*|evidence_rest)
      # (evidence_rest: everything but the rocprofv3 passes -- after those have been distilled on the CPU side, so that the bench
      # lines carry the fresh counters' tag: tools/gpu_profile_all.sh first, tools/distill_profile.py, then this)
      export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
      tag=${R#evidence:}
      [ "$R" != evidence_rest ] && bash tools/gpu_profile_all.sh $tag c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3 > $O/profile_all.log 2>&1
      "$0" $TAG bench_all force_dist pf node_gather pf_consumer sums_time
      for wl in c3 c5 c4; do bash tools/gpu_pmc_branch.sh $wl > /dev/null 2>&1; cp gpurun_out/pmc_branch_$wl.txt $O/; done
      for wl in c4 c2x16 c10k; do for who in dev host; do timeout 600 python3 tools/e2e_breakdown.py $wl $who 2>&1 | grep -v amdgpu > $O/e2e_${wl}_$who.txt; done; done
      timeout 900 python3 bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json ;;
    sums_time)
      { timeout 300 python3 tools/sums_time.py 10240 1 5 f64; timeout 300 python3 tools/sums_time.py 1024 32 5 f64; timeout 300 python3 tools/sums_time.py 40960 1 5 f64
        timeout 300 python3 tools/sums_time.py 65536 1 5 f32; timeout 300 python3 tools/sums_time.py 32768 1 5 f32; timeout 300 python3 tools/sums_time.py 16384 1 5 f32; } 2>&1 | grep -v amdgpu.ids > $O/sums_time.txt
      cat $O/sums_time.txt ;;
    fuzz_sums)
      export FUZZ_OUT=$O
      run() { name=$1; shift; env "$@" timeout 1500 python3 tools/fuzz_gpu.py $N $SEED > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log; echo "== $name: $(grep -c '^trial' $O/$name.log) trials, $(tail -1 $O/$name.log)"; grep -i 'mismatch\|error\|assert' $O/$name.log | head -3; }
      N=900 SEED=100708 run coop_sums FUZZ_COOP=1 FUZZ_SUMS=1
      N=400 SEED=100709 run ragged_coop_sums FUZZ_COOP=1 FUZZ_SUMS=1 FUZZ_RAGGED=1
      N=500 SEED=100710 run opt_sums FUZZ_R5=1 FUZZ_OPT=1 FUZZ_SUMS=1
      N=500 SEED=100711 run ncyc_sums FUZZ_R5=1 FUZZ_NCYC=1 FUZZ_SUMS=1
      grep -h "sums(" $O/*_sums.log | grep -o "sums(k=[0-9]*,[0-9]* launches,[A-Za-z]*" | sed 's/.*,//' | sort | uniq -c ;;
    fuzz)
      FUZZ_OUT=$O/fuzz bash tools/gpu_fuzz.sh ${FUZZ_ARGS:-600 400 300 600 150} ;;
    *) echo "unknown recipe $R" ;;
  esac
done
