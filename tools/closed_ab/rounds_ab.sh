for r in 4 3 2; do for fit in 0.75 0.9; do
  for mode in "--hbm-only" ""; do
    PBSIM_JOB_ROUNDS=$r PBSIM_JOB_FIT=$fit timeout 500 python bench.py $mode --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['critical_path']['per_rank'][0]
print('rounds=$r fit=$fit mode=${mode:-delivered}', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],1), 'ms | launches', d['roofline']['launches'], 'walk', round(r['walk'],1))"
  done; done; done
