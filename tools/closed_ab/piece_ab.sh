#!/bin/bash
# GPU box, EXPERIMENTAL build: 32-KiB chunks per deflate piece (PBSIM_DEFLATE_PIECE_CHUNKS; 8192 = 256 MiB of text ships) for
# ranks 0 / 3 / 7 of the eight-rank configs[1] job (rounds of ~1 GB of text per lane) and for the one-GPU job
cd "$(dirname "$0")/../.."
one() { python3 bench.py --no-cpu-baseline --no-extras --steps 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f ms per step, pcie_frac %.3f' % (j['ms_per_step'], j['delivery']['pcie_frac']))"; }
ranks() { PBSIM_REPLAY_ONLY=0,3,7 python3 bench.py --replay-ranks 8 --c1-gbs 0 --no-extras --no-cpu-baseline --steps 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(', '.join('rank %d of 8: %.1f ms (busy %.1f)' % (x['rank'], x['wall_ms'], x['breakdown_ms']['worker_busy']) for x in d['replay']['by_world']['8']['per_rank']))"; }
for pc in 8192 4096 2048 1024 8192; do
  export PBSIM_DEFLATE_PIECE_CHUNKS=$pc
  echo "piece chunks $pc: $(ranks)"
done
for pc in 8192 4096 2048; do
  export PBSIM_DEFLATE_PIECE_CHUNKS=$pc
  echo "piece chunks $pc: one GPU $(one)"
done
