#!/bin/bash
# GPU box: whole-job wall time (text left in HBM) over the job pipeline's knobs: rounds per record x rounds in flight
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rounds in 2 3 4; do
  for depth in 2 3; do
    echo "== PBSIM_JOB_ROUNDS=$rounds PBSIM_JOB_DEPTH=$depth"
    PBSIM_JOB_ROUNDS=$rounds PBSIM_JOB_DEPTH=$depth python bench.py --no-cpu-baseline --no-extras --hbm-only --steps 2 2>/dev/null |
      python -c "import json,sys; j=json.load(sys.stdin); print('value %.1f Gbases/s  ms_per_step %.1f  rounds %d  walk_busy %.0f ms  avg_launch %.1f ms' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['roofline']['walk_busy_ms'], j['roofline']['avg_launch_ms']))"
  done
done
