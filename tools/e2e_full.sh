#!/bin/bash
# usage (GPU box, repo root): tools/e2e_full.sh [RECORDS] [RECORD_BP] [DEPTH] [MODEL] [EXTRA CLI ARGS...]
# End to end at the BASELINE size: a FASTA of RECORDS x RECORD_BP uniform ACGT in /dev/shm -> the pbsim CLI
# -> <prefix>_NNNN.fq.gz + .maf.gz (+ .ref) in /dev/shm, wall time by phase (PBSIM_TRACE=1: the CLI's phase clock and the job's).
nrec=${1:-4}; bp=${2:-750000000}; depth=${3:-20}; model=${4:-ERRHMM-ONT.model}
shift; shift; shift; shift
d=$(mktemp -d /dev/shm/pbsim_e2e.XXXX)
python3 - $nrec $bp $d <<'PY'
import sys, time, numpy as np
nrec, n, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
t0 = time.time()
with open(d + "/g.fa", "wb") as f:
    for r in range(nrec):
        rng = np.random.default_rng(100 + r)
        f.write(b">chr%d synthetic\n" % (r + 1))
        for a in range(0, n, 80_000_000):
            m = min(80_000_000, n - a)
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, m, dtype=np.uint8)].reshape(-1, 80)
            f.write(np.concatenate([s, np.full((s.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
print("FASTA written in %.1f s" % (time.time() - t0), file=sys.stderr)
PY
M=$(python3 -c "import sys; sys.path.insert(0,'tests'); import harness; print(harness.model_path('$model'))")
ls -la $d/g.fa | awk '{print "FASTA bytes", $5}'
free -g | head -2
for rep in 1 2; do
  t0=$(date +%s.%N)
  PBSIM_TRACE=1 pbsim3_amd/bin/pbsim --strategy wgs --method errhmm --errhmm $M --genome $d/g.fa --depth $depth --seed 1 --prefix $d/out "$@" 2> $d/err.txt
  rc=$?
  t1=$(date +%s.%N)
  echo "== run $rep rc=$rc wall $(python3 -c "print(round($t1 - $t0, 2))") s"
  grep "pbsim cli\]" $d/err.txt
  grep "worker: bytes of rec" $d/err.txt | awk '{s+=$(NF-5)} END {print "delivery worker busy (sum over rounds): " s " ms"}'
  grep -c "pbsim job r0\] t=.* round first" $d/err.txt | sed 's/^/rounds: /'
  grep "read num\|^depth" $d/err.txt | head -8 | tr "\n" " "; echo
  du -sb $d/out_*.gz 2>/dev/null | awk '{s+=$1} END {print "compressed output bytes", s}'
  du -sb $d/out_*.ref 2>/dev/null | awk '{s+=$1} END {print ".ref bytes", s}'
  rm -f $d/out_*
done
rm -rf $d
