#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/qshmm_ranks_ab.sh -- ranks 0 and 7 of the eight-rank configs[2] job (QSHMM-RSII --pass-num 10, one
# 750 Mbp record) replayed alone on the GPU: the default lane / wave split of the rounds against the lane walker only
one() { PBSIM_REPLAY_ONLY=0,7 python3 bench.py --workload qshmm10 --records 1 --no-cpu-baseline --no-extras --replay-ranks 8 --c1-gbs 0 --steps 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=j['replay']['by_world']['8']
print('one GPU %.1f ms; ' % j['ms_per_step'] + ', '.join('rank %d of 8: %.1f ms (walk wait %.1f, bytes wait %.1f, rounds %d)' % (x['rank'], x['wall_ms'], x['breakdown_ms']['wait_walk'], x['breakdown_ms']['wait_bytes'], x['rounds']) for x in w['per_rank']))"; }
echo -n "default split: "; one
echo -n "lane walker only (PBSIM_COOP_LEN=-1): "; PBSIM_COOP_LEN=-1 one
echo -n "every task by waves up to 60 000, else lanes (rounds 3-4's rule, PBSIM_COOP_LEN by hand: n/a) -- wave walker alone, 200 000 tasks: "; PBSIM_COOP_LEN=0 python3 tools/walk_solo.py qshmm10 3 20000 2>&1 | tail -1
for n in 6000 10000 20000 40000; do echo -n "default split, $n reads x 10: "; python3 tools/walk_solo.py qshmm10 3 $n 2>&1 | tail -1; done
