#!/bin/bash
# A/B of job knobs on the replayed N-rank job (tools/replay_ranks.py): ranks 0, 3 and N-1 of 8, both BASELINE configs
# usage: tools/replay_sweep.sh OUT.txt "ENV1=a ENV2=b" "ENV1=c" ...
out=$1; shift
: > $out
for wl in errhmm onthq60; do
  for kv in "$@"; do
    echo "== $wl $kv" >> $out
    env PBSIM_REPLAY_ONLY=0,3,7 $kv python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --workload $wl --replay-ranks 8 2>> $out.err | python -c '
import json,sys
d=json.loads(sys.stdin.readline())
r=d["replay"]["by_world"]["8"]
print("  t1_ms %.1f" % d["replay"]["t1_ms"])
for x in r["per_rank"]:
    b=x["breakdown_ms"]
    print("  rank %d wall %.1f (less callbacks %.1f) GB %.2f  wait_walk %.0f wait_bytes %.0f merge %.0f tail %.0f worker_busy %.0f" % (x["rank"], x["wall_ms"], x["wall_less_callbacks_ms"], x["host_bytes"]/1e9, b["wait_walk"], b["wait_bytes"], b["merge"], b["tail_block"]+b["drain"], b["worker_busy"]))
' >> $out
  done
done
cat $out
