#!/bin/bash
# GPU box: the delivery-inclusive job over rounds in flight (PBSIM_JOB_DEPTH)
cd "$(dirname "$0")/../.."
for depth in 1 2 3; do
  echo "== PBSIM_JOB_DEPTH=$depth"
  PBSIM_JOB_DEPTH=$depth python bench.py --no-cpu-baseline --no-extras --steps 2 2>/dev/null |
    python -c "import json,sys; j=json.load(sys.stdin); r=j['roofline']; print('value %.1f Gbases/s  ms_per_step %.1f  rounds %d  pcie %.2f | walk achieved %.0f GB/s frac %.3f avg_launch %.1f ms x%d  tails %s' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['delivery']['pcie_frac'], r['achieved'], r['frac'], r['avg_launch_ms'], r['launches'], r['tail_read_launches']))"
done
