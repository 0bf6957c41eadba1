#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/deflate_timeline.sh OUT  -- kernel trace of one delivered job: how long the full-size
# k_deflate_chunks launches (8192 chunks) take by what runs beside them (a walk kernel, the other lane's deflate kernel)
out=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out
rocprofv3 --kernel-trace --output-format csv -d $R/$out/t -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 "$@" > $R/$out/bench.json 2> $R/$out/bench.err
cd $R
python3 - $out <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
rows=[]
for f in glob.glob(out+"/t/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
rows.sort()
walks=[(a,b) for a,b,n,g in rows if "k_walk" in n and b-a>1e6]
dfl=[(a,b) for a,b,n,g in rows if "k_deflate_chunks" in n and g>=8192*256]
texts=[(a,b) for a,b,n,g in rows if "k_text_rows" in n]
def overlap(a,b,ivs):
    t=0
    for x,y in ivs:
        if y<=a or x>=b: continue
        t+=min(b,y)-max(a,x)
    return t/(b-a)
cls=collections.defaultdict(list)
for a,b in dfl:
    w=overlap(a,b,walks); o=overlap(a,b,[(x,y) for x,y in dfl if (x,y)!=(a,b)]); t=overlap(a,b,texts)
    key=("walk" if w>0.5 else "no walk", "other lane" if o>0.5 else "alone", "text" if t>0.3 else "no text")
    cls[key].append((b-a)/1e6)
print("full-size k_deflate_chunks launches: %d" % len(dfl))
for k,v in sorted(cls.items()):
    v.sort()
    print("  %-34s n=%4d  median %.2f ms  mean %.2f  p10 %.2f  p90 %.2f" % (" + ".join(k), len(v), v[len(v)//2], sum(v)/len(v), v[len(v)//10], v[len(v)*9//10]))
PY
