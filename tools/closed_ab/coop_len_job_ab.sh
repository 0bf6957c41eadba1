#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_len_job_ab.sh  -- the configs[1] job by the read length from which the wave walker
# takes the reads of its full rounds (PBSIM_COOP_LEN; the rule gives 4 x the mean length = 36096 for them), in HBM and delivered
for t in 27136 36096 45056 54272 63488; do
  for rep in 1 2; do
    PBSIM_COOP_LEN=$t python bench.py --hbm-only --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('job in HBM, wave walker from $t: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))"
  done
done
for t in 27136 36096 45056 54272; do
  PBSIM_COOP_LEN=$t python bench.py --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('job delivered, wave walker from $t: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))"
done
