for cfg in "41 default" "41 -1" "85 -1" "85 default" "60 -1" "100 -1"; do
  set -- $cfg
  export PBSIM_WALK_LDS_KB=$1
  if [ $2 = default ]; then unset PBSIM_COOP_LEN; else export PBSIM_COOP_LEN=$2; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk lds $1 coop $2: value', round(d['value']/1e9,1), 'ms', round(d['ms_per_step'],1), 'pcie', round(d['delivery']['pcie_frac'],3), 'walk ms', round(d['roofline']['avg_launch_ms'],1))"
done
