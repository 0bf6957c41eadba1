#!/bin/bash
# usage (GPU box, repo root): tools/host_contention.sh TAG [RANKS_TO_REPLAY] [RATE_GBS]
# Eight ranks' load on the host's memory, on the one-GPU box: ranks of the eight-rank configs[1] job are
# replayed alone on the GPU (bench.py --replay-ranks 8) twice -- on the quiet box, and beside tools/host_load.c, which stands in
# for the seven other ranks' 47 GB/s of DMA writes into their NUMA nodes + the sinks reading them back.
tag=$1; only=${2:-0,3,7}; rate=${3:-47}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
cc -O2 -pthread -o /tmp/host_load $R/tools/host_load.c || exit 1
node=$(python3 -c "
import re, pbsim3_amd as P
m = re.search(r'numa node (\d+)', str(P.bind_host_to_device(0) or ''))
print(m.group(1) if m else 0)")
echo "real rank's GPU on NUMA node $node; $(ls -d /sys/devices/system/node/node* | wc -l) node(s), $(nproc) cpus" > $out/${tag}_host_contention.log
run() {  # run NAME
  PBSIM_REPLAY_ONLY=$only python3 bench.py --replay-ranks 8 --no-extras --no-cpu-baseline --steps 2 > $out/${tag}_contention_$1.json 2> $out/${tag}_contention_$1.err
}
run quiet
/tmp/host_load 8 $rate 600 0 $node > $out/${tag}_host_load.log 2>&1 &
hl=$!
sleep 2
run loaded
kill -INT $hl 2>/dev/null; sleep 1; kill $hl 2>/dev/null
wait $hl 2>/dev/null
# twice the rate: where does it start to hurt?
/tmp/host_load 8 $((rate * 2)) 600 0 $node > $out/${tag}_host_load2.log 2>&1 &
hl=$!
sleep 2
run loaded2
kill $hl 2>/dev/null
wait $hl 2>/dev/null
python3 - $out $tag $only <<'PY' | tee -a $out/${tag}_host_contention.log
import json, sys
out, tag, only = sys.argv[1:4]
def load(name):
    d = json.loads(open(f"{out}/{tag}_contention_{name}.json").read().strip().splitlines()[-1])
    w = d["replay"]["by_world"]["8"]
    return d["ms_per_step"], {x["rank"]: x for x in w["per_rank"]}
rows = {n: load(n) for n in ("quiet", "loaded", "loaded2")}
print("one-GPU job (ms per step):", {n: round(v[0], 1) for n, v in rows.items()})
for r in sorted(rows["quiet"][1]):
    q, l, l2 = (rows[n][1][r] for n in ("quiet", "loaded", "loaded2"))
    print("rank %d of 8: wall %.1f -> %.1f -> %.1f ms; delivery thread busy %.1f -> %.1f -> %.1f ms; waited for bytes %.1f -> %.1f -> %.1f ms" % (
        r, q["wall_ms"], l["wall_ms"], l2["wall_ms"], q["breakdown_ms"]["worker_busy"], l["breakdown_ms"]["worker_busy"], l2["breakdown_ms"]["worker_busy"],
        q["breakdown_ms"]["wait_bytes"], l["breakdown_ms"]["wait_bytes"], l2["breakdown_ms"]["wait_bytes"]))
PY
cat $out/${tag}_host_load.log $out/${tag}_host_load2.log >> $out/${tag}_host_contention.log
