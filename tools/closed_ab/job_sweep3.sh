#!/bin/bash
# GPU box: whole-job wall time (text left in HBM) with the wave walker in place: rounds per record x rounds in flight x the
# lane walk's workgroups per CU (PBSIM_WALK_LDS_KB 41 = three, 0 = what fits)
cd "$(dirname "$0")/../.."
for kb in 41 0; do for rounds in 4 6 8; do for depth in 2 3; do
  PBSIM_WALK_LDS_KB=$kb PBSIM_JOB_ROUNDS=$rounds PBSIM_JOB_DEPTH=$depth python bench.py --no-cpu-baseline --no-extras --hbm-only --steps 2 2>/dev/null |
    python -c "import json,sys; j=json.load(sys.stdin); print('lds $kb rounds/record $rounds depth $depth: %.1f Gbases/s  %.1f ms  rounds %d  avg_launch %.1f ms' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['roofline']['avg_launch_ms']))"
done; done; done
