#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_wg_ab.sh  -- the wave walker's persistent workgroups (default 1024 = 4 per CU):
# alone (every read of a 100 000-read and a 300 000-read batch), and in the configs[1] job in HBM and delivered
for wg in 1024 1280 1536 2048; do
  for n in 100000 300000; do
    PBSIM_COOP_WG=$wg PBSIM_COOP_LEN=0 python tools/walk_solo.py errhmm 3 $n 2>/dev/null | grep -v JSON | sed "s/^/wave only, $wg workgroups: /"
  done
done
PBSIM_COOP_WG=1280 PBSIM_COOP_LEN=0 python tools/walk_solo.py onthq 3 100000 2>/dev/null | grep -v JSON | sed "s/^/wave only, 1280 workgroups: /"
PBSIM_COOP_WG=1024 PBSIM_COOP_LEN=0 python tools/walk_solo.py onthq 3 100000 2>/dev/null | grep -v JSON | sed "s/^/wave only, 1024 workgroups: /"
for wg in 1024 1280 2048; do
  for rep in 1 2; do
    PBSIM_COOP_WG=$wg python bench.py --hbm-only --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('job in HBM, $wg workgroups: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))"
  done
done
for wg in 1024 1280 2048; do
  PBSIM_COOP_WG=$wg python bench.py --no-extras --steps 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('job delivered, $wg workgroups: %.1f ms  %.1f Gbases/s' % (d['ms_per_step'], d['value'] / 1e9))"
done
