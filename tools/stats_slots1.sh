#!/bin/bash
# usage (GPU box): tools/stats_slots1.sh OUTDIR [bench args] -- per-kernel times with one slot (no overlap between kernels)
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -- python3 $R/bench.py --no-cpu-baseline --slots 1 "$@" > $R/$out/bench.json 2> $R/$out/err.txt
cd $R
python3 - $out <<'PY'
import csv,glob,sys,json
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
d=json.loads([l for l in open(sys.argv[1]+"/bench.json") if l.startswith("{")][-1])
print("value %.1f Gbases/s, ms/step %.1f, bases/step %.2f G" % (d["value"]/1e9, d["ms_per_step"], d["config"]["bases_per_step"]/1e9))
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-50s calls %3s avg %9.3f ms" % (r["Name"].replace("pbsim::(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e6))
PY
