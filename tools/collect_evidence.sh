#!/bin/bash
# after `gpurun -- 'tools/gpu_lease.sh <dir> evidence:<tag>'`: distil what came back under gpurun_out/ into profiles/ (tracked)
#   usage: tools/collect_evidence.sh <dir> <tag>      e.g. tools/collect_evidence.sh r6ev r06
cd "$(dirname "$0")/.." || exit 1
E=gpurun_out/$1; T=$2
for wl in c10k c2 c2x16 c3 c4 c5 c10kn c4n c10kr3; do python3 tools/distill_profile.py gpurun_out/prof_${T}_$wl > /dev/null || echo "distill $wl failed"; done
cp $E/bench_all.jsonl profiles/${T}_bench_all.jsonl
cp $E/bench_default.json profiles/${T}_bench_default.json
python3 - "$E" "$T" <<'PY'
import json, sys
E, T = sys.argv[1:3]
rows = [json.loads(l) for l in open('%s/force_dist.jsonl' % E) if l.strip().startswith('{')]
json.dump(rows, open('profiles/%s_force_dist.json' % T, 'w'), indent=1)
print(len(rows), 'force-dist lines')
PY
{ echo "# the particle filter's cycle from a C99 host (tests/c/pf_consumer.c: 131 072 particles x 48 steps + analysis, 300 cycles);"
  echo "# one_wave = 1: node and twin forced onto the one-wave kernel (two shards of 65 536 would take the four-chunk cooperative kernel by"
  echo "# shape, the twin's 131 072 the one-wave kernel: equal to rounding, not to bits -- state_max_rel_diff says how equal)"
  for f in $E/pf_consumer_*.log; do echo "## $(basename $f .log)"; cat $f; done; } > profiles/${T}_pf_consumer.txt
for wl in c3 c5 c4; do cp $E/pmc_branch_$wl.txt profiles/${T}_${wl}_instruction_mix.txt; done
{ echo "# whole job, one forcing alone: tools/e2e_breakdown.py <workload> dev|host"; for wl in c4 c2x16 c10k; do for who in dev host; do echo "== $wl, plans by the $who"; cut -c1-260 $E/e2e_${wl}_$who.txt; done; done; } > profiles/${T}_e2e_breakdown.txt
{ echo "# sipnet_batch_pf_resample_peers at the slot counts of 1 / 2 / 4 / 8 ranks on one GPU (tools/pf_peers_time.py), then the kernels' own"
  echo "# durations at 8 x 131 072 slots (rocprofv3 --kernel-trace --stats of the same tool, both parameter modes)"
  cat $E/pf_peers_time.txt; echo; head -4 $E/pf_peers_w8_kernel_stats.csv | cut -c1-300; } > profiles/${T}_pf_peers_time.txt
{ echo "# tools/node_gather_time.py at c10k's shape: what the member-resolved exchange costs the C host (one shard / two shards on one GPU)"; for d in 0 0,0; do echo "## devices $d"; cat $E/node_gather_time_$d.txt; done; } > profiles/${T}_node_gather_time.txt
python3 - "$T" <<'PY'
import json, sys
T = sys.argv[1]
for l in open('profiles/%s_bench_all.jsonl' % T):
    d = json.loads(l); r = d['roofline']; e = r.get('end_to_end') or {}
    print(d['config']['workload'][:30], 'ms', round(d['ms_per_step'], 4), 'frac', round(r['frac'], 3), 'traffic_tag', str(r.get('traffic_tag'))[:30], 'e2e', e.get('ms') and round(e['ms'], 2), e.get('pipelined_ms') and round(e['pipelined_ms'], 2))
PY
