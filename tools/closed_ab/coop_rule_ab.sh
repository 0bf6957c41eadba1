#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_rule_ab.sh OUT  -- the lane / wave split rule (reads of one mean length and more go to
# the wave walker at PBSIM_COOP_SPLIT_READS reads per batch: 150 000 = round 2's rule, 80 000 = the default): the configs[1]
# job in HBM and delivered, ranks 0 / 3 / 7 of eight replayed
out=$1
: > $out
RULES="PBSIM_COOP_SPLIT_READS=150000 PBSIM_COOP_SPLIT_READS=110000 PBSIM_COOP_SPLIT_READS=80000 "
for rep in 1 2; do
  for kv in $RULES; do
    env $kv python bench.py --hbm-only --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$kv job in HBM: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))" >> $out
  done
done
for rep in 1 2; do
  for kv in $RULES; do
    env $kv python bench.py --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$kv job delivered: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))" >> $out
  done
done
tools/replay_sweep.sh $out.replay $RULES > /dev/null 2>&1
cat $out.replay >> $out
cat $out
