for cfg in "16 -1" "16 default" "4 -1" "4 default"; do
  set -- $cfg
  export GPU_MAX_HW_QUEUES=$1
  if [ $2 = default ]; then unset PBSIM_COOP_LEN; else export PBSIM_COOP_LEN=$2; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $1 coop $2: value', round(d['value']/1e9,1), 'hbm', round(d['whole_job_hbm']['value']/1e9,1), 'steady', round(d['steady_state_hbm']['value']/1e9,1), 'walk ms', round(d['roofline']['avg_launch_ms'],1), 'tail ms', round(d['roofline']['tail_read_launches']['avg_ms'],2))"
done
