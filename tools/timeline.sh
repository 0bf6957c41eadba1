#!/bin/bash
# usage (GPU box): tools/timeline.sh OUTDIR [bench args] -- kernel timeline (start, duration, queue) of the last 2 bench steps
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$out -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $R/$out/bench.json 2> $R/$out/err.txt
cd $R
python3 - $out <<'PY'
import csv,glob,sys
f=max(glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True), key=lambda p: __import__("os").path.getsize(p))   # the parent's (the largest), not a child's
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t_end=max(int(r["End_Timestamp"]) for r in rows)
sel=[r for r in rows if t_end-330e6 < int(r["Start_Timestamp"]) < t_end-130e6]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d<0.15: continue
    name=r["Kernel_Name"].replace("void ","").replace("pbsim::(anonymous namespace)::","").split("(")[0][:28]
    print("%8.2f ms  +%7.2f ms  q%-3s %s" % ((int(r["Start_Timestamp"])-t0)/1e6, d, r.get("Queue_Id","?"), name))
PY
