#!/bin/bash
# GPU box, EXPERIMENTAL build (PBSIM_EXTRA_CFLAGS=-DPBSIM_EXPERIMENTAL): whole job with its text left in HBM over rounds per
# record x rounds in flight x lane / wave split
cd "$(dirname "$0")/../.."
for coop in default -1; do
for rounds in 1 2 3 4; do
  for depth in 2 3; do
    if [ $coop = default ]; then unset PBSIM_COOP_LEN; else export PBSIM_COOP_LEN=$coop; fi
    echo -n "coop=$coop rounds=$rounds depth=$depth: "
    PBSIM_JOB_ROUNDS=$rounds PBSIM_JOB_DEPTH=$depth python bench.py --no-cpu-baseline --no-extras --hbm-only --steps 2 2>/dev/null |
      python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f Gbases/s  ms_per_step %.1f  rounds %d  walk_busy %.0f ms  avg_launch %.1f ms' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['roofline']['walk_busy_ms'], j['roofline']['avg_launch_ms']))"
  done
done
done
