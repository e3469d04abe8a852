#!/bin/bash
# The first run on a node with more than one GPU, as ONE command.  Everything lands under gpurun_out/scale/ and the final
# table in profiles/scale_table.md (+ scale_lines.jsonl): copy nothing by hand.
#   usage: tools/scale_run.sh [max_gpus]   (default: every visible GPU, at most 8)
# What it runs, in this order (each leg under its own timeout; a leg that fails is reported in the table, the rest go on):
#   1. the two-real-GPU tests that one-GPU boxes skip (member and site shards, overlapped gathers, peer-read filter cycles)
#   2. `python3 bench.py --gpus N` for N = 1, 2, 4, 8 -- bench.py starts its own ranks (torch.distributed.run as a child;
#      what the driver's SCALE run does) -- for c10k, c4, c5: the contract's lines (statistics exchange / peer-read filter)
#   3. the north star's exchange as written: `--gather full` (the member-resolved planes, one all-gather after the pass)
#      and its segmented-overlap measurement (config.gather_full of the default line) at N = max; `--gather sums` (every
#      member's daily sums out of the step kernel's own launch, gathered under the next pass) at every N
#   4. c5 with `--pf-exchange alltoall` at N = max (the fallback for ranks that cannot map each other's HBM)
#   5. the C host (sipnet_node_*: one process, one thread + one RCCL rank per GPU): node_consumer (statistics + planes +
#      the reduced member-resolved gather), pf_consumer (config 5's cycle), at N = 1, 2, 4, 8 with per-GPU work fixed
#   5b. the C host with a process per GPU (tests/c/rank_consumer.c: sipnet_comm_* among N processes) at N = 1, 2, 4, 8
#   6. the CLI: `sipnet --sites LIST --devices 0-(N-1)` over 8 x N run directories (whole sites per device)
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
MAX=${1:-$HAVE}; [ "$MAX" -gt 8 ] && MAX=8
O=gpurun_out/scale; mkdir -p $O profiles
: > $O/scale_lines.jsonl
echo "visible GPUs: $HAVE, running up to $MAX ($(date -u +%FT%TZ))" | tee $O/README.txt
NS=""; for n in 1 2 4 8; do [ "$n" -le "$MAX" ] && NS="$NS $n"; done

# 1. tests that need two real GPUs
if [ "$HAVE" -ge 2 ]; then
  timeout 1800 python3 -m pytest tests/test_gpu_node.py tests/test_gpu_multirank.py -q -m gpu -k "two_real or real_devices" > $O/pytest_two_gpus.txt 2>&1
  grep -E "passed|failed|error" $O/pytest_two_gpus.txt | tail -1 | tee -a $O/README.txt
fi

# 2. the bench contract's lines (bench.py launches its own ranks)
line() {   # line <tag> <bench args...>: one bench run, its JSON line tagged and appended
  local tag=$1; shift
  timeout 1500 python3 bench.py "$@" > $O/bench_$tag.log 2>&1
  local rc=$?
  local j=$(grep '^{' $O/bench_$tag.log | tail -1)
  if [ -n "$j" ]; then echo "$j" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d['leg']='$tag'; d['rc']=$rc; print(json.dumps(d))" >> $O/scale_lines.jsonl
  else echo "{\"leg\": \"$tag\", \"rc\": $rc, \"failed\": true}" >> $O/scale_lines.jsonl; fi
  echo "$tag rc=$rc"
}
for wl in c10k c4 c5; do
  steps=20; warm=3; [ "$wl" = c5 ] && steps=400 && warm=40
  for n in $NS; do
    extra=""; [ "$n" -eq 1 ] && extra="--no-cpu-baseline"
    line ${wl}_n$n --workload $wl --gpus $n --steps $steps --warmup $warm $extra
  done
done
# 3. the member-resolved planes, gathered after the pass (the default line's config.gather_full is the overlapped variant)
[ "$MAX" -ge 2 ] && for wl in c10k c4; do line ${wl}_full_n$MAX --workload $wl --gpus $MAX --steps 10 --warmup 2 --gather full; done
# 3b. the member-resolved exchange that fits under the kernel: every member's daily sums from the step kernel's own launch,
#     all-gathered under the next pass (90 MB per rank and year at c10k) -- at every N, for a curve of its own
for wl in c10k c4; do for n in $NS; do [ "$n" -ge 2 ] && line ${wl}_sums_n$n --workload $wl --gpus $n --steps 20 --warmup 3 --gather sums; done; done
# 4. the filter's all-to-all fallback, and its peer exchange with the all-gather through torch.distributed's process group
#    (the default line takes the engine's own RCCL communicator on the batch's stream)
[ "$MAX" -ge 2 ] && line c5_alltoall_n$MAX --workload c5 --gpus $MAX --steps 400 --warmup 40 --pf-exchange alltoall
[ "$MAX" -ge 2 ] && line c5_torchcoll_n$MAX --workload c5 --gpus $MAX --steps 400 --warmup 40 --pf-collective torch

# 5. the C host
gcc -std=c99 -O1 -Iinclude tests/c/node_consumer.c -o /tmp/node_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
gcc -std=c99 -O1 -Iinclude tests/c/pf_consumer.c -o /tmp/pf_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from sipnet_amd import synth
synth.write_clim("/tmp/year.clim", synth.round_like_file(synth.half_hourly_year_raw(17520)))
synth.write_clim("/tmp/day.clim", synth.round_like_file(synth.half_hourly_year_raw(48)))
PY
for n in $NS; do
  devs=$(seq -s, 0 $((n - 1)))
  # weak scaling: 10 240 members / 131 072 particles PER GPU
  timeout 900 /tmp/node_consumer sipnet_amd/data/base_forest.param /tmp/year.clim $((10240 * n)) $devs > $O/node_consumer_n$n.log 2>&1
  echo "rc=$?" >> $O/node_consumer_n$n.log
  timeout 900 /tmp/pf_consumer sipnet_amd/data/base_forest.param /tmp/day.clim $((131072 * n)) $devs 300 48 > $O/pf_consumer_n$n.log 2>&1
  echo "rc=$?" >> $O/pf_consumer_n$n.log
  timeout 900 python3 tools/node_gather_time.py 10240 1 $devs > $O/node_gather_time_n$n.txt 2>&1     # plain / overlapped planes / daily sums / fp32
done

# 5b. the C host with a PROCESS per GPU (tests/c/rank_consumer.c: the batch API, in-launch daily sums, ONE all-gather through the
#     engine's own RCCL communicator -- ncclCommInitRank among N processes, the id through a file)
gcc -std=c99 -O1 -Iinclude tests/c/rank_consumer.c -o /tmp/rank_consumer -Lsipnet_amd -lsipnet_amd -Wl,-rpath,$PWD/sipnet_amd
for n in $NS; do
  rm -f /tmp/comm_n$n.id
  for r in $(seq 0 $((n - 1))); do
    timeout 900 /tmp/rank_consumer sipnet_amd/data/base_forest.param /tmp/year.clim $((10240 * n)) $n $r $r /tmp/comm_n$n.id > $O/rank_consumer_n${n}_r$r.log 2>&1 &
  done
  wait
  cp $O/rank_consumer_n${n}_r0.log $O/rank_consumer_n$n.log; echo "ranks_ok=$(grep -l '^rc=0' $O/rank_consumer_n${n}_r*.log | wc -l)" >> $O/rank_consumer_n$n.log
done

# 6. the CLI over run directories, whole sites per device
python3 tools/cli_sites_time.py --sites $((8 * MAX)) --devices 0-$((MAX - 1)) > $O/cli_sites_n$MAX.txt 2>&1 || echo "cli leg rc=$?" >> $O/cli_sites_n$MAX.txt

# the table
python3 - "$O" <<'PY'
import json, os, sys
O = sys.argv[1]
rows = [json.loads(l) for l in open(os.path.join(O, "scale_lines.jsonl"))]
base = {}
out = ["| leg | N | value (%s) | ms / pass | efficiency vs N = 1 | ranks_seen | devices_seen | rc | exchange per pass | planes gathered, overlapped (ms) | daily sums gathered, overlapped (ms) |" % "ensemble-site-timesteps/s",
       "|---|---|---|---|---|---|---|---|---|---|---|"]
for d in rows:
    if d.get("failed"):
        out.append("| %s | | FAILED | | | | | %s |" % (d["leg"], d["rc"]))
        continue
    wl = d["leg"].split("_")[0]
    n = d["n_gpus"]
    if d["leg"] == "%s_n1" % wl:
        base[wl] = d["value"]
    eff = d["value"] / (n * base[wl]) if wl in base else float("nan")
    c = d["config"]
    pfc = c.get("particle_filter") or {}
    ms = lambda k: ("%.2f" % c[k]["ms"]) if (c.get(k) or {}).get("ms") else ("failed" if c.get(k) else "")
    out.append("| %s | %d | %.4g | %.4f | %.3f | %s | %s | %s | %s | %s | %s |" % (
        d["leg"], n, d["value"], d["ms_per_step"], eff, c.get("ranks_seen"), c.get("devices_seen"), d["rc"],
        ("%s, %s" % (pfc.get("exchange"), "engine's communicator" if "sipnet_comm" in str(pfc.get("collective")) else "torch process group")) if pfc.get("exchange") and not str(pfc.get("exchange")).startswith("n/a")
        else (pfc.get("exchange") or c.get("gather", "")), ms("gather_full"), ms("gather_sums")))
for n in (1, 2, 4, 8):
    for f in ("node_consumer", "pf_consumer", "rank_consumer"):
        p = os.path.join(O, "%s_n%d.log" % (f, n))
        if os.path.exists(p):
            kv = dict(l.strip().split("=", 1) for l in open(p) if "=" in l)
            keys = [k for k in kv if k.startswith("ms_") or k in ("rc", "state_identical", "reduced_sums_equal_planes", "collective_library", "comm_world", "ranks_ok", "kernel")]
            out.append("| C host: %s | %d | %s |" % (f, n, ", ".join("%s=%s" % (k, kv[k]) for k in keys)))
open("profiles/scale_table.md", "w").write("\n".join(out) + "\n")
open("profiles/scale_lines.jsonl", "w").write(open(os.path.join(O, "scale_lines.jsonl")).read())
print("\n".join(out))
PY
cp profiles/scale_table.md $O/ 2>/dev/null
ls $O
