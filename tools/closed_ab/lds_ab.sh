#!/bin/bash
# GPU box: same-box A/B of the walk's workgroups-per-CU cap (PBSIM_WALK_LDS_KB): solo walk, steady state, whole job in HBM
cd "$(dirname "$0")/../.."
for rep in 1 2; do for kb in 0 27 41; do
  echo "== rep $rep PBSIM_WALK_LDS_KB=$kb"
  PBSIM_WALK_LDS_KB=$kb python tools/walk_solo.py errhmm 3 2>/dev/null | tail -1
  PBSIM_WALK_LDS_KB=$kb python bench.py --no-cpu-baseline --hbm-only --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); s=j['steady_state_hbm']; print('  hbm job %.1f G/s %.0f ms | steady %.1f G/s walk launch %.1f ms' % (j['value']/1e9, j['ms_per_step'], s['value']/1e9, s['walk']['avg_launch_ms']))"
done; done
