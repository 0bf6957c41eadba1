#!/bin/bash
# usage (GPU box, repo root): tools/trace_ab.sh TAG -- how much does rocprofv3 --kernel-trace perturb the delivered job?
# The same command plain and under the tracer (and both with HSA_ENABLE_INTERRUPT=0: signals polled instead of interrupt-driven),
# each run's own line (ms_per_step, the walk's HIP-event avg_launch_ms) beside the CSV's average for the walk kernel.
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
ARGS="--no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail ''"
cd /tmp && export TMPDIR=/tmp
run() {  # run NAME traced(0|1) [ENV=VAL ..]
  local name=$1 traced=$2; shift; shift
  for kv in "$@"; do export "$kv"; done
  if [ $traced = 1 ]; then
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail "" > $out/$name.json 2> $out/$name.err
    cp $(find $out/$name -name "*kernel_stats.csv" | head -1) $out/$name.kernel_stats.csv 2>/dev/null
    rm -rf $out/$name
  else
    timeout 600 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail "" > $out/$name.json 2> $out/$name.err
  fi
  for kv in "$@"; do unset "${kv%%=*}"; done
  grep '^{' $out/$name.json | tail -1 > $out/$name.line; mv $out/$name.line $out/$name.json
}
run plain 0
run traced 1
run plain_poll 0 HSA_ENABLE_INTERRUPT=0
run traced_poll 1 HSA_ENABLE_INTERRUPT=0
cd $R
python3 - $out <<'PY'
import csv, json, os, sys
d = sys.argv[1]
for name in ("plain", "traced", "plain_poll", "traced_poll"):
    try:
        j = json.load(open(os.path.join(d, name + ".json")))
    except Exception as e:
        print(name, "no line:", e); continue
    rf = j["roofline"]
    row = "%-12s ms/step %8.1f  walk HIP-event avg %7.3f ms x %d  frac %.4f" % (name, j["ms_per_step"], rf["avg_launch_ms"], rf["launches"], rf["frac"])
    p = os.path.join(d, name + ".kernel_stats.csv")
    if os.path.exists(p):
        calls = tot = 0
        for r in csv.DictReader(open(p)):
            if "k_walk_errhmm<" in r["Name"] or r["Name"].split("(")[0].endswith("k_walk_errhmm"):
                calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])
        if calls:
            avg = tot / calls / 1e6
            row += "  | CSV avg %7.3f ms x %d  frac %.4f" % (avg, calls, rf["alg_bytes_per_launch"] / (avg / 1e3) / 1e9 / 8000)
    print(row)
PY
