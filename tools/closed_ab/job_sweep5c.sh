#!/bin/bash
# GPU box, EXPERIMENTAL build: the HBM-only job at two rounds per record, three in flight: ramp, wave-walker workgroups
cd "$(dirname "$0")/../.."
one() { python bench.py --no-cpu-baseline --no-extras "$@" --steps 3 2>/dev/null |
      python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f Gbases/s  ms_per_step %.1f  rounds %d  walk_busy %.0f ms  avg_launch %.1f ms' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['roofline']['walk_busy_ms'], j['roofline']['avg_launch_ms']))"; }
export PBSIM_JOB_ROUNDS=2 PBSIM_JOB_DEPTH=3
echo -n "base: "; one --hbm-only
echo -n "ramp off: "; PBSIM_JOB_RAMP=0 one --hbm-only
for wg in 256 384 768 1024; do echo -n "coop wg $wg: "; PBSIM_COOP_WG=$wg one --hbm-only; done
echo -n "coop dynamic 1: "; PBSIM_COOP_DYNAMIC=1 one --hbm-only
echo -n "base again: "; one --hbm-only
echo -n "split 60000: "; PBSIM_COOP_SPLIT_READS=60000 one --hbm-only
echo -n "split 110000: "; PBSIM_COOP_SPLIT_READS=110000 one --hbm-only
