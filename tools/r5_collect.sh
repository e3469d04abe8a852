#!/bin/bash
# after `gpurun -- bash tools/gpu_r5_evidence.sh`: distil what came back under gpurun_out/ into profiles/ (tracked)
cd "$(dirname "$0")/.." || exit 1
E=gpurun_out/r5ev
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do python tools/distill_profile.py gpurun_out/prof_r05_$wl > /dev/null || echo "distill $wl failed"; done
cp $E/bench_all.jsonl profiles/r05_bench_all.jsonl
python - <<'PY'
import json
rows = [json.loads(l) for l in open('gpurun_out/r5ev/force_dist.jsonl') if l.strip().startswith('{')]
json.dump(rows, open('profiles/r05_force_dist.json', 'w'), indent=1)
print(len(rows), 'force-dist lines')
PY
{ echo "# the particle filter's cycle from a C99 host (tests/c/pf_consumer.c: 131 072 particles x 48 steps + analysis, 300 cycles)"; for d in 0 0,0; do echo "## devices $d"; cat $E/pf_consumer_$d.log; done; } > profiles/r05_pf_consumer.txt
for wl in c3 c5 c4; do cp $E/pmc_branch_$wl.txt profiles/r05_${wl}_instruction_mix.txt; done
{ echo "# whole job, one forcing alone: tools/e2e_breakdown.py <workload> dev|host (phase by phase with a synchronisation after each phase,"; echo "# then without any: the host side until the step kernel is queued, and the total)"; for wl in c4 c2x16 c10k; do for who in dev host; do echo "== $wl, plans by the $who"; cat $E/e2e_${wl}_$who.txt | cut -c1-260; done; done; } > profiles/r05_e2e_breakdown.txt
cp $E/cli_block_time.txt profiles/r05_cli_block_time.txt
{ echo "# round 5: site plans built on the device (csrc/plan_device.h) -- MI355X, tools/gpu_r5_evidence.sh"; echo; echo "## the ring walk per forcing shape (tools/plan_device_time.py: clock stamps of planSeqKernel under SIPNET_KOPT_DEVICE_PLAN; every record"; echo "## and eviction compared with the host builder's bytes)"; cat $E/plan_device_time.txt; echo; echo "-> one lane walks ~0.45-0.6 us a step (a lone wavefront issues an instruction every ~6 cycles); a run of equal step lengths is"; echo "   walked for 5 days / length + a few steps, the rest is one descriptor.  Default policy: at most 1 024 walked steps (kDevPlanMaxWalked),"; echo "   else the host builds (niwot).  The year-to-date GDD chain measured 203 us per 17 520 records on the device (v_add_f64 dependent"; echo "   issue) against ~25 us on a host core: it is the plan threads' (plan.cpp buildSitePlanLight)."; echo; echo "## kernels of one hand-over at c4 (32 sites x 17 520 records; rocprofv3 --kernel-trace --stats of tools/e2e_breakdown.py c4)"; grep -i "plan\|Name\|setupKernel\|convertParams" $E/plan_kernel_stats_c4.csv; echo; echo "## history of planSeqKernel at c4: 457 us (wave-serial loops of global loads) -> 215 (parallel parts in the wide kernels) -> 121 (GDD chain to the host)"; echo "## whole-job numbers: profiles/r05_e2e_breakdown.txt, profiles/r05_bench_all.jsonl (end_to_end.ms / pipelined_ms / plan_device_sites)"; } > profiles/r05_plan_device.txt
tail -3 $E/pytest_gpu.txt
python - <<'PY'
import json
for l in open('profiles/r05_bench_all.jsonl'):
    d = json.loads(l); r = d['roofline']; e = r.get('end_to_end') or {}
    print(d['config']['workload'][:30], 'ms', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 3), 'traffic_tag', str(r.get('traffic_tag'))[:30], 'e2e', e.get('ms') and round(e['ms'], 2), e.get('pipelined_ms') and round(e['pipelined_ms'], 2))
PY
