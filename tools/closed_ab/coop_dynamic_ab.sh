for v in 0 1 0 1; do
  for mode in "--hbm-only" ""; do
    PBSIM_COOP_DYNAMIC=$v timeout 500 python bench.py $mode --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['critical_path']['per_rank'][0]
print('dynamic=$v mode=${mode:-delivered}', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],1), 'ms | walk', round(r['walk'],1))"
  done
done
for v in 0 1; do echo "solo coop-all 50k reads dynamic=$v"; PBSIM_COOP_DYNAMIC=$v PBSIM_COOP_LEN=0 timeout 200 python tools/walk_solo.py errhmm 5 50000 2>&1 | tail -1; done
