#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_walk.sh OUTDIR [bench args...]
# two PMC passes (SQ has 8 slots per pass) over one bench step; prints per-wave-step figures for k_walk_*
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out/p1 $out/p2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/p1 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $out/bench1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d $out/p2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $out/bench2.log 2>&1
python3 - $out <<'PY'
import csv,glob,collections,sys,json
out=sys.argv[1]
agg=collections.defaultdict(float)
disp=collections.defaultdict(set)
for f in glob.glob(out+"/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_walk" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in agg: agg[k]/=max(1,len(disp[k]))   # average per launch
d=json.loads([l for l in open(out+"/bench1.log") if l.startswith("{")][-1])
steps=d["config"]["bases_per_step"]*1.062/64
print("wave-steps %.3g" % steps)
for k,v in sorted(agg.items()): print("%-24s %.4g   per wave-step %.1f" % (k,v,v/steps))
PY
