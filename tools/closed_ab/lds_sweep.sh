#!/bin/bash
# GPU box: walk workgroups per CU (PBSIM_WALK_LDS_KB) vs solo walk speed, whole job in HBM, whole job delivered
cd "$(dirname "$0")/../.."
for kb in 0 27 33 41; do
  echo "== PBSIM_WALK_LDS_KB=$kb"
  PBSIM_WALK_LDS_KB=$kb python tools/walk_solo.py errhmm 2 2>/dev/null | tail -1
  PBSIM_WALK_LDS_KB=$kb python bench.py --no-cpu-baseline --no-extras --hbm-only --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('  hbm job %.1f G/s %.0f ms rounds %d' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step']))"
  PBSIM_WALK_LDS_KB=$kb python bench.py --no-cpu-baseline --no-extras --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('  delivered job %.1f G/s %.0f ms' % (j['value']/1e9, j['ms_per_step']))"
done
