#!/bin/bash
# usage (GPU box): tools/e2e_cli.sh GENOME_BP [MODES]  -- end-to-end wall time of the pbsim CLI (FASTA on disk -> files on disk)
# MODES: comma list of plain,gpu,host (default all three; host = zlib on the CPU threads, minutes at 1 Gbp)
n=${1:-100000000}
d=$(mktemp -d /dev/shm/pbsim_e2e.XXXX)
python3 - $n $d <<'PY'
import sys, numpy as np
n=int(sys.argv[1]); d=sys.argv[2]
rng=np.random.default_rng(1)
s=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,n)].reshape(-1,80)
out=np.concatenate([s,np.full((s.shape[0],1),10,np.uint8)],axis=1)
open(d+"/g.fa","wb").write(b">chr1\n"+out.tobytes())
PY
python3 -c "import sys; sys.path.insert(0,'tests'); import harness; print(harness.model_path('ERRHMM-ONT.model'))" > $d/model.txt
M=$(cat $d/model.txt)
modes=${2:-plain,gpu,host}
for m in ${modes//,/ }; do
  case $m in plain) mode="--no-gzip";; gpu) mode="";; host) mode="--gzip host";; esac
  t0=$(date +%s.%N)
  pbsim3_amd/bin/pbsim --strategy wgs --method errhmm --errhmm $M --genome $d/g.fa --depth 20 --seed 1 --prefix $d/out $mode 2> $d/err.txt
  t1=$(date +%s.%N)
  echo "mode=[$mode] wall $(python3 -c "print(round($t1 - $t0, 2))") s; cpu/elapsed: $(grep -A2 "System utilization" $d/err.txt | tail -2 | tr "\n" " ")"; grep "read num\|depth :" $d/err.txt | tail -2; ls -la $d | grep out_0001 | awk '{print $5, $9}'
  rm -f $d/out_*
done
rm -rf $d
