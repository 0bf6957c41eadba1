#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/pmc_deflate.sh OUTDIR -- PMC picture of k_deflate_chunks on FASTQ-/MAF-like text
out=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out/p1 $out/p2
export NOPROF=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p1 -- python3 tools/deflate_prof.py > $out/log1.txt 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/p2 -- python3 tools/deflate_prof.py > $out/log2.txt 2>&1
python3 - $out <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
rows=collections.defaultdict(dict)
for f in glob.glob(out+"/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_deflate_chunks" in r["Kernel_Name"]:
            rows[(r["Grid_Size"])].setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for g,c in sorted(rows.items(), key=lambda x:int(x[0])):
    nch=int(g)//256
    if nch < 100: continue
    print("grid",g,"chunks",nch)
    for k,v in sorted(c.items()):
        print("  %-24s per chunk %.0f   (%d launches)" % (k, sum(v)/len(v)/nch, len(v)))
PY
