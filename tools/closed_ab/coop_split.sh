#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_split.sh [model]  -- one walk launch at a time, by batch size and by the length from
# which reads go to the wave walker (PBSIM_COOP_LEN; -1 = lane walker only, 0 = wave walker only)
m=${1:-errhmm}
for n in 50000 100000 200000 450000; do
  for t in -1 36096 27136 18176 13824 9216 4608 0; do
    PBSIM_COOP_LEN=$t python tools/walk_solo.py $m 3 $n 2>/dev/null | awk -v n=$n -v t=$t '{print n" reads, from "t": "$6" ms"}'
  done
done
