#!/bin/bash
# usage (GPU box, repo root): tools/timeline_job.sh OUT [bench args]  -- kernel timeline of the last job run of bench.py:
# per kernel name the summed duration, and how much of the wall time had 0 / 1 / 2+ walk kernels running
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
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
# the last job = after the last long gap? take the final 45 % of the trace span as "the timed step" approximation: find walk launches
walks=[r for r in rows if "k_walk" in r[2]]
t_end=max(r[1] for r in rows)
# timed step starts at the first walk launch after the biggest idle gap between walk activity
# bench.py runs the job twice (one warm-up, one timed step): the timed step begins at the first round of record 1 of the
# second run = the first big-batch walk launch in the second half of the launches' count
big=[w for w in walks if w[1]-w[0] > 2e6]
start=big[len(big)//2][0]
sel=[r for r in rows if r[0]>=start]
span=(max(r[1] for r in sel)-start)/1e6
print("timed step: %.1f ms, %d kernel launches" % (span,len(sel)))
agg=collections.defaultdict(float); cnt=collections.Counter()
for a,b,n in sel:
    import re
    m=re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", n)
    k=m.group(1) if m else n[:40]
    agg[k]+= (b-a)/1e6; cnt[k]+=1
for k,v in sorted(agg.items(),key=lambda x:-x[1])[:12]:
    print("  %-40s %8.1f ms  x%d" % (k,v,cnt[k]))
# concurrency of walk kernels over time
ev=[]
for a,b,n in sel:
    if "k_walk" in n: ev+= [(a,1),(b,-1)]
ev.sort()
cur=0; last=start; hist=collections.defaultdict(float)
for t,d in ev:
    hist[min(cur,3)]+= (t-last)/1e6; last=t; cur+=d
hist[0]+= (max(r[1] for r in sel)-last)/1e6
print("  walk kernels running: " + ", ".join("%d%s: %.1f ms" % (k,"+" if k==3 else "",v) for k,v in sorted(hist.items())))
# any-kernel busy
ev=[]
for a,b,n in sel: ev+=[(a,1),(b,-1)]
ev.sort(); cur=0; last=start; idle=0
for t,d in ev:
    if cur==0: idle+=(t-last)/1e6
    last=t; cur+=d
print("  no kernel at all running: %.1f ms" % idle)
PY
rm -rf $out/t
