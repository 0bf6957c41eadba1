#!/bin/bash
# usage (GPU box): tools/prof_cli.sh OUTDIR GENOME_BP [cli flags...] -- rocprofv3 kernel stats of one pbsim CLI run
out=$1; n=$2; shift; shift
d=$(mktemp -d /dev/shm/pbsim_prof.XXXX)
python3 - $n $d <<'PY'
import sys, numpy as np
n=int(sys.argv[1]); d=sys.argv[2]
rng=np.random.default_rng(1)
s=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,n)].reshape(-1,80)
out=np.concatenate([s,np.full((s.shape[0],1),10,np.uint8)],axis=1)
open(d+"/g.fa","wb").write(b">chr1\n"+out.tobytes())
PY
M=$(python3 -c "import sys; sys.path.insert(0,'tests'); import harness; print(harness.model_path('ERRHMM-ONT.model'))")
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
# (an ordinary exit: the profiler writes its files from an exit handler, which pbsim's default _exit() would skip)
export PBSIM_CLI_LEAVE_CONTEXT=0
PBSIM_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -- $R/pbsim3_amd/bin/pbsim --strategy wgs --method errhmm --errhmm $M --genome $d/g.fa --depth 20 --seed 1 --prefix $d/out "$@" 2> $R/$out/err.txt
cd $R
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s total %10.2f ms avg %9.3f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
grep "trace\]" $out/err.txt | head -20
rm -rf $d
