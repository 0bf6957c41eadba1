#!/bin/bash
# usage (GPU box): tools/e2e_sample.sh N_STRINGS GENOME_BP -- end-to-end wall time of `pbsim --method sample`: a sample FASTQ and a
# FASTA on disk (/dev/shm) -> .fq.gz + .maf.gz on disk, with PBSIM_TRACE-free timing of the whole process
n=${1:-50000}; g=${2:-25000000}
d=$(mktemp -d /dev/shm/pbsim_e2es.XXXX)
python3 - $n $g $d <<'PY'
import sys, numpy as np
n=int(sys.argv[1]); g=int(sys.argv[2]); d=sys.argv[3]
rng=np.random.default_rng(1)
s=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,g)].reshape(-1,80)
out=np.concatenate([s,np.full((s.shape[0],1),10,np.uint8)],axis=1)
open(d+"/g.fa","wb").write(b">chr1\n"+out.tobytes())
k=(9000.0/7000.0)**2
lens=np.clip(rng.gamma(k,9000.0/k,n),100,60000).astype(np.int64)
level=rng.integers(8,31,n)
with open(d+"/s.fastq","wb") as f:
    for i in range(n):
        L=int(lens[i])
        q=(np.clip(level[i]+rng.integers(-5,6,L),0,93).astype(np.uint8)+33).tobytes()
        f.write(b"@r%d\n"%i + b"A"*L + b"\n+\n" + q + b"\n")
print("fastq bytes", int(lens.sum())*2)
PY
for mode in "" "--no-gzip"; do
  t0=$(date +%s.%N)
  PBSIM_TRACE=1 pbsim3_amd/bin/pbsim --strategy wgs --method sample --sample $d/s.fastq --genome $d/g.fa --depth 20 --seed 1 --prefix $d/out $mode 2> $d/err.txt
  rc=$?
  t1=$(date +%s.%N)
  grep "pbsim cli" $d/err.txt; echo "mode=[$mode] rc=$rc wall $(python3 -c "print(round($t1 - $t0, 2))") s"; grep "read num\|depth :" $d/err.txt | tail -2; ls -la $d | grep out_0001 | awk '{print $5, $9}'
  rm -f $d/out_* sample_profile_*
done
rm -rf $d
