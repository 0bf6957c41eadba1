#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_solo.sh [model] [reads]  -- one walk launch at a time: lane walker alone, wave
# walker alone (every read) by the number of its persistent workgroups, and the default split
m=${1:-errhmm}; n=${2:-100000}
PBSIM_COOP_LEN=-1 python tools/walk_solo.py $m 3 $n 2>/dev/null | sed 's/^/lane only: /'
for wg in 256 512 1024 2048; do PBSIM_COOP_WG=$wg PBSIM_COOP_LEN=0 python tools/walk_solo.py $m 3 $n 2>/dev/null | sed "s/^/wave only, $wg workgroups: /"; done
python tools/walk_solo.py $m 3 $n 2>/dev/null | sed 's/^/default split: /'
