#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh OUTDIR KERNEL_SUBSTRING [walk_solo args...]
# two SQ passes + FETCH/WRITE passes (each counter set in its own run, --kernel-trace only) over tools/walk_solo.py: one walk
# batch at a time, nothing overlapping; per-launch averages for the named kernel
out=$1; k=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out/p1 $R/$out/p2 $R/$out/p3 $R/$out/p4
B="python3 $R/tools/walk_solo.py $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$out/p1 -- $B > $R/$out/log1.txt 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/$out/p2 -- $B > $R/$out/log2.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/p3 -- $B > $R/$out/log3.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/p4 -- $B > $R/$out/log4.txt 2>&1
cd $R
tail -1 $out/log1.txt
python3 - $out "$k" <<'PY'
import csv,glob,collections,sys
out,k=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(out+"/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out+"/p1/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("kernel", k, "launches", len(dur), "avg ms %.2f" % (sum(dur)/max(1,len(dur))))
for n,v in sorted(agg.items()):
    a=sum(v)/len(v)
    extra=" = %.2f GB" % (a*1024/1e9) if n in ("FETCH_SIZE","WRITE_SIZE") else ""
    print("  %-22s %.4g%s" % (n,a,extra))
PY
