#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_split_ab.sh  -- the lane / wave split after the wave walker's step was cut: the batch
# size at which reads of one mean length and more go to the wave walker (PBSIM_COOP_SPLIT_READS, default 150 000), one walk
# launch at a time by batch size, then the configs[1] job in HBM and rank 0 / 3 / 7 of eight replayed
for n in 100000 200000 450000; do
  for sr in 150000 200000 260000 340000; do
    PBSIM_COOP_SPLIT_READS=$sr python tools/walk_solo.py errhmm 3 $n 2>/dev/null | awk -v n=$n -v t=$sr '{print n" reads, split at "t": "$6" ms"}'
  done
done
for sr in 150000 200000 260000; do
  for rep in 1 2; do
    PBSIM_COOP_SPLIT_READS=$sr python bench.py --hbm-only --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('job in HBM, split at $sr: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))"
  done
done
