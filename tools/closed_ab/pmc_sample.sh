#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/pmc_sample.sh OUTDIR -- SQ / FETCH / WRITE counters of the sampling method's kernels over
# tools/sample_prof.py 60000 (two runs of the 2-Gbase job: four k_walk_sample launches, the two long ones are the first sweeps);
# each counter set in its own run with --kernel-trace only
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out/p1 $R/$out/p2 $R/$out/p3 $R/$out/p4
B="python3 $R/tools/sample_prof.py 60000"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$out/p1 -- $B > $R/$out/log1.txt 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/$out/p2 -- $B > $R/$out/log2.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/p3 -- $B > $R/$out/log3.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/p4 -- $B > $R/$out/log4.txt 2>&1
cd $R
grep clip $out/log1.txt | tail -1
python3 - $out <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for k in ("k_walk_sample","k_sample_qsum","k_text_rows"):
    per=collections.defaultdict(dict)   # (pass dir, dispatch id) -> counters
    dur={}
    for f in glob.glob(out+"/p*/**/*counter_collection.csv",recursive=True):
        p=f.split("/p")[1][0]
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]: per[(p,r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    for f in glob.glob(out+"/p*/**/*kernel_trace.csv",recursive=True):
        p=f.split("/p")[1][0]
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]: dur[(p,r["Dispatch_Id"])]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    # the longest launch of each pass = a first sweep
    best={}
    for (p,d),ms in dur.items():
        if p not in best or ms>dur[(p,best[p])]: best[p]=d
    print("kernel",k)
    for p in sorted(best):
        c=per.get((p,best[p]),{})
        print("  pass",p,"longest launch %.2f ms"%dur[(p,best[p])], {n:("%.4g"%v if n not in ("FETCH_SIZE","WRITE_SIZE") else "%.2f GB"%(v*1024/1e9)) for n,v in sorted(c.items())})
PY
