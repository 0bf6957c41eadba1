#!/bin/bash
# GPU box, EXPERIMENTAL build: follow-up of job_sweep5.sh
cd "$(dirname "$0")/../.."
one() { python bench.py --no-cpu-baseline --no-extras "$@" --steps 2 2>/dev/null |
      python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1f Gbases/s  ms_per_step %.1f  rounds %d  walk_busy %.0f ms  avg_launch %.1f ms pcie %.3f' % (j['value']/1e9, j['ms_per_step'], j['config']['rounds_per_step'], j['roofline']['walk_busy_ms'], j['roofline']['avg_launch_ms'], j['delivery']['pcie_frac']))"; }
for i in 1 2 3; do echo -n "hbm rounds=2 depth=3 (#$i): "; PBSIM_JOB_ROUNDS=2 PBSIM_JOB_DEPTH=3 one --hbm-only; done
for i in 1 2; do echo -n "hbm rounds=4 depth=3 (#$i): "; PBSIM_JOB_ROUNDS=4 PBSIM_JOB_DEPTH=3 one --hbm-only; done
for sr in 40000 160000 320000; do echo -n "hbm rounds=2 depth=3 split_reads=$sr: "; PBSIM_COOP_SPLIT_READS=$sr PBSIM_JOB_ROUNDS=2 PBSIM_JOB_DEPTH=3 one --hbm-only; done
for r in 2 3 4; do echo -n "delivered rounds=$r: "; PBSIM_JOB_ROUNDS=$r one; done
for r in 2 4; do echo -n "delivered onthq60 rounds=$r: "; PBSIM_JOB_ROUNDS=$r one --workload onthq60; done
for r in 2 4; do echo -n "hbm onthq60 rounds=$r depth=3: "; PBSIM_JOB_ROUNDS=$r PBSIM_JOB_DEPTH=3 one --hbm-only --workload onthq60; done
for r in 2 4; do echo -n "hbm qshmm10 1 record rounds=$r depth=3: "; PBSIM_JOB_ROUNDS=$r PBSIM_JOB_DEPTH=3 one --hbm-only --workload qshmm10 --records 1; done
